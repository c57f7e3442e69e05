"""Interleaved A/B of library builds in ONE process launch sequence: python tools_ab.py lib1.so lib2.so ..."""
import subprocess, sys, json, os
libs = sys.argv[1:]
res = {l: [] for l in libs}
for rnd in range(3):
    for l in libs:
        env = dict(os.environ, NSR_LIB_PATH=os.path.abspath(l))
        out = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe_network_vs_render.py"), "render_only"], env=env, capture_output=True, text=True).stdout
        ms = [json.loads(x)["render_ms"] for x in out.splitlines() if "render_ms" in x]
        res[l].append(ms[-1])
for l in libs:
    print(os.path.basename(l), ["%.2f" % x for x in res[l]], "min %.2f" % min(res[l]))
