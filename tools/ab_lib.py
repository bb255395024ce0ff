#!/usr/bin/env python3
"""A/B timing of alternative builds of the library: python tools/ab_lib.py libA.so libB.so ... (one process each)"""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for lib in sys.argv[1:]:
    env = dict(os.environ, R2S_LIB_OVERRIDE=os.path.abspath(lib))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "profile_modes.py"), "--modes", "sdf", "--reps", os.environ.get("AB_REPS", "100")],
                         env=env, capture_output=True, text=True)
    line = [l for l in out.stdout.splitlines() if l.startswith("sdf")]
    if line:
        d = json.loads(line[-1].split(" ", 1)[1])
        print(os.path.basename(lib), {k: v for k, v in d.items() if k.startswith("ms_")})
    else:
        print(os.path.basename(lib), out.stderr[-400:])
