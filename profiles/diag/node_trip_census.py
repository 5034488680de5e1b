import json, os, subprocess, sys
for w, spp in (("hcandersen", 256), ("dragon", 128), ("synthetic", 32), ("teapot", 256)):
    env = dict(os.environ, PTAMD_LIB=os.path.join(os.getcwd(), "cuda-path-tracer_amd/lib/variants/libptamd_census.so"))
    r = subprocess.run([sys.executable, "bench.py", "--workload", w, "--spp", str(spp), "--steps", "1", "--warmup", "0", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=300)
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    s = d["schedule"]; t = s["node"]["trips"]
    print(w, "per node trip: walking %.1f  finished %.1f  at a leaf %.1f  idle/entity %.1f" % (s["node"]["lanes"] / t, s["ent"]["lanes"] / t, s["leaf"]["lanes"] / t, s["front"]["lanes"] / t), flush=True)
