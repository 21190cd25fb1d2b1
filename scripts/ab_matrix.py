"""A/B of several builds of libhpt.so and/or tuning variants on one box: every (library, scene) cell is measured by
scripts/ab_tuning.py in its own process, AB_ROUNDS rounds, the order of the libraries rotated from round to round (the
later process of a group runs a little faster whatever it loads).  Prints the median of each cell's medians.
usage: python scripts/ab_matrix.py <lib name or 'default'> [...]     env: AB_* of ab_tuning.py, ABM_ROUNDS (default 3)
a lib name X means path_tracing_amd/csrc/libhpt_X.so (see `make variant`)."""
import os, re, subprocess, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
libs = sys.argv[1:] or ["default"]
rounds = int(os.environ.get("ABM_ROUNDS", "3"))
pat = re.compile(r"variant\s+(\S+) render ([\d.]+) ms \(min ([\d.]+)\) \| first ([\d.]+) resume ([\d.]+) shade ([\d.]+) other ([\d.]+)")
res = {}
head = None
for r in range(rounds):
    order = libs[r % len(libs):] + libs[:r % len(libs)]
    for lib in order:
        env = dict(os.environ)
        env.setdefault("AB_ROUNDS", "3")
        if lib != "default": env["HPT_LIBRARY"] = os.path.join(ROOT, "path_tracing_amd", "csrc", "libhpt_%s.so" % lib)
        else: env.pop("HPT_LIBRARY", None)
        out = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "ab_tuning.py")], env=env, capture_output=True, text=True)
        if out.returncode != 0:
            print("lib %s failed:\n%s" % (lib, out.stderr[-2000:]), flush=True); continue
        for line in out.stdout.splitlines():
            if line.startswith("scene"): head = line
            m = pat.search(line)
            if m: res.setdefault((lib, m.group(1)), []).append([float(m.group(i)) for i in (2, 4, 5, 6, 7)])
print(head)
for (lib, var), v in res.items():
    m = np.median(np.array(v), axis=0)
    print("%-10s %-14s render %7.2f | first %6.2f resume %6.2f shade %6.2f other %5.2f   (%d runs: %s)" % (
        lib, var, m[0], m[1], m[2], m[3], m[4], len(v), " ".join("%.2f" % x[0] for x in v)), flush=True)
