"""Turn the rocprofv3 sqlite outputs under gpurun_out/prof/ (tools/profile_r04.sh) into small text summaries in profiles/."""
import json, os, sqlite3, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(ROOT, "gpurun_out", "prof")
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
# "box": run on the GPU box right after the passes -- summaries go to gpurun_out/prof_txt (merged back), databases are deleted
on_box = len(sys.argv) > 2 and sys.argv[2] == "box"
out = os.path.join(ROOT, "gpurun_out", "prof_txt") if on_box else os.path.join(ROOT, "profiles")
os.makedirs(out, exist_ok=True)

WHAT = {"cg": "python bench.py --steps 100 --warmup 10 (512^3 CG+Jacobi, the headline)",
        "gmres": "python bench.py --solver gmres --precond ilu0 --steps 60 --warmup 10 (512^3 GMRES(30)+ILU(0))",
        "shell": "python bench.py --matrix shell --solver gmres --precond ilu0 --steps 60 --warmup 10 (config 3 surrogate)",
        "shell_rcm": "python bench.py --matrix shell --shell-variant rcm --solver gmres --precond ilu0 --steps 60 --warmup 10 (the config-3 class in "
                     "reverse Cuthill-McKee order: the sync-free grouped triangular solve, k_trsv_sf)",
        "bicgstab": "RAMD_MC_RB=0 python bench.py --solver bicgstab --precond mcsgs --steps 60 --warmup 10 (512^3 BiCGStab+MC-SGS, config 4's solver; "
                    "the colour sweeps, as before the red-black form of round 5)",
        "ell": "RAMD_MC_RB=0 python bench.py --format ell --solver bicgstab --precond mcsgs --steps 60 --warmup 10 (config 4: ELL interior; colour sweeps)",
        "hyb": "RAMD_MC_RB=0 python bench.py --format hyb --solver bicgstab --precond mcsgs --steps 60 --warmup 10 (config 4: HYB interior; colour sweeps)",
        "bicgstab_rb": "python bench.py --solver bicgstab --precond mcsgs --steps 60 --warmup 10 (512^3 BiCGStab+MC-SGS, the default: k_mc_rb)",
        "ell_rb": "python bench.py --format ell --solver bicgstab --precond mcsgs --steps 60 --warmup 10 (config 4: ELL interior, the default: k_mc_rb)",
        "hyb_rb": "python bench.py --format hyb --solver bicgstab --precond mcsgs --steps 60 --warmup 10 (config 4: HYB interior, the default: k_mc_rb)",
        "mixed": "python bench.py --solver mixed --steps 30 --warmup 3 (config 5: fp64 defect correction around fp32 CG+Jacobi)",
        "lap27_cg": "python bench.py --matrix lap27 --grid 256 --steps 100 --warmup 10 (the reference's own 3-D operator, 27-point, 256^3: CG+Jacobi)",
        "lap27_gmres": "python bench.py --matrix lap27 --grid 256 --solver gmres --precond ilu0 --steps 60 --warmup 10 (27-point 256^3: the pencil "
                       "triangular solve k_trsv_box)",
        "lap27_bicgstab": "python bench.py --matrix lap27 --grid 256 --solver bicgstab --precond mcsgs --steps 60 --warmup 10 (27-point 256^3: 8 colours)",
        "lap27_ell": "python bench.py --matrix lap27 --grid 256 --format ell --steps 60 --warmup 10 (27-point 256^3, ELL with row patterns)",
        "lap27_hyb": "python bench.py --matrix lap27 --grid 256 --format hyb --steps 60 --warmup 10 (27-point 256^3, HYB with row patterns)",
        "calib": "tools/_bin/membench calib (reads of 1 GiB with 16 / 8 / 4 bytes per lane, of 256 MiB with 1 byte per lane)"}


def q(db, sql):
    return sqlite3.connect(db).cursor().execute(sql).fetchall()


def short(name):
    return name.replace("void ramd::", "ramd::")[:120]


for name, what in WHAT.items():
    kt = os.path.join(src, "kt_" + name, "bench_results.db")
    if not os.path.exists(kt):
        continue
    with open(os.path.join(out, "%s_kernel_stats_%s.txt" % (tag, name)), "w") as f:
        f.write("# rocprofv3 --kernel-trace --stats -- %s, MI355X\n" % what)
        f.write("# name | calls | total_us | avg_us | pct\n")
        for r in q(kt, "select name,total_calls,total_duration,average,percentage from top_kernels order by total_duration desc limit 30"):
            f.write("%s | %d | %.1f | %.3f | %.2f\n" % (short(r[0]), r[1], r[2] / 1.0, r[3], r[4]))
        bj = os.path.join(src, "bench_kt_%s.json" % name)
        if os.path.exists(bj):
            f.write("# bench line of the same (profiled) run:\n# " + open(bj).read().strip() + "\n")

traffic = {}
for name in WHAT:
    vals = {}
    if name == "calib" and tag == "r02":
        continue
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        db = os.path.join(src, "%s_%s" % (ctr, name), "bench_results.db")
        if not os.path.exists(db):
            continue
        rows = q(db, "select kernel_name,counter_name,count(*),avg(value),min(value),max(value) from counters_collection "
                     "group by kernel_name,counter_name order by avg(value) desc limit 48")
        with open(os.path.join(out, "%s_pmc_%s_%s.txt" % (tag, ctr, name)), "w") as f:
            f.write("# rocprofv3 --pmc %s --kernel-trace -- %s   (one counter per pass)\n" % (ctr, WHAT[name].split(" (")[0].replace("--steps 100 --warmup 10", "--steps 20 --warmup 2").replace("--steps 60 --warmup 10", "--steps 20 --warmup 2").replace("--steps 30 --warmup 3", "--steps 10 --warmup 2")))
            f.write("# KiB per dispatch as reported.  gfx950: FETCH_SIZE counts 64 B per 128-B request, i.e. HALF the bytes read\n"
                    "# (MI355X_MICROARCH.md, HBM section) -- calibrated in this round on reads of known size with 16, 8, 4 and 1 bytes per\n"
                    "# lane (r03_pmc_FETCH_SIZE_calib.txt: 0.5000 of the bytes every time; round 2's 4-byte kernel had been optimised away and\n"
                    "# its conclusion that narrow reads are counted in full was wrong).  WRITE_SIZE matches the written volume (calib: a 1-GiB\n"
                    "# fill reports 1 GiB).  HBM bytes = 2 x FETCH_SIZE + WRITE_SIZE.\n")
            f.write("# kernel | counter | dispatches | avg | min | max\n")
            for r in rows:
                f.write("%s | %s | %d | %.1f | %.1f | %.1f\n" % (short(r[0]), r[1], r[2], r[3], r[4], r[5]))
        for r in rows:
            vals.setdefault(short(r[0]), {})[ctr] = (r[3], r[2])
    traffic[name] = vals
raw_path = os.path.join(out, tag + "_pmc_raw.json")
if os.path.exists(raw_path):  # a pass over some legs only keeps the other legs' figures
    old = json.load(open(raw_path))
    for k, v in old.items():
        if not traffic.get(k):
            traffic[k] = v
traffic = {k: v for k, v in traffic.items() if v}
json.dump(traffic, open(raw_path, "w"), indent=1)
if on_box:
    import shutil
    for d in os.listdir(src):
        if os.path.isdir(os.path.join(src, d)):
            shutil.rmtree(os.path.join(src, d), ignore_errors=True)
print(sorted(os.listdir(out)))
