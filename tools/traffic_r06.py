"""profiles/r06_pmc_raw.json (per-kernel FETCH_SIZE / WRITE_SIZE averages of tools/profile_r06.sh) -> profiles/r06_traffic.json,
the table bench.py's `traffic_for` reads.  bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (MI355X_MICROARCH.md, HBM section:
both counters in KiB, FETCH_SIZE counts 64 B per 128-B request on gfx950).  Keys of kernels this round did not touch are carried over
from profiles/r05_traffic.json where round 6 has no pass of its own."""
import json
import os
import sys

here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
raw_path = os.path.join(here, "r06_pmc_raw.json")
later = os.path.join(here, "..", "gpurun_out", "prof_txt", "r06_pmc_raw.json")
raw = json.load(open(raw_path)) if os.path.exists(raw_path) else {}
if os.path.exists(later):
    for leg, v in json.load(open(later)).items():
        if v:
            raw[leg] = v
    json.dump(raw, open(raw_path, "w"), indent=1)
t = {k: v for k, v in json.load(open(os.path.join(here, "r05_traffic.json"))).items() if k not in ("how", "notes")}
t["how"] = __doc__.replace("\n", " ")
fresh = []


def find(leg, *needles):
    hits = [(k, v) for k, v in raw.get(leg, {}).items() if all(s in k for s in needles) and "FETCH_SIZE" in v and "WRITE_SIZE" in v]
    if len(hits) != 1:
        print("%s %r: %d kernels match -- key left as it was" % (leg, needles, len(hits)))
        return None
    v = hits[0][1]
    return int(round((2 * v["FETCH_SIZE"][0] + v["WRITE_SIZE"][0]) * 1024))


def put(key, leg, *needles):
    v = find(leg, *needles)
    if v is not None:
        t[key] = v
        fresh.append(key)


n, nnz = 512 ** 3, 7 * 512 ** 3 - 6 * 512 ** 2
put("spmv_csr_512", "cg", "k_csr_pat2<double, 0, true")
put("spmv_csr_512_columns_read", "cg", "k_csr_tr<double, 0, true")
put("cg_update_512", "cg", "k_cg_update<double")
put("trsv_512_lower", "gmres", "k_trsv_lat<double, true")
put("trsv_512_upper", "gmres", "k_trsv_lat<double, false")
if "trsv_512_lower" in fresh and "trsv_512_upper" in fresh:
    t["trsv_512"] = (t["trsv_512_lower"] + t["trsv_512_upper"]) // 2
    t["lusolve_512"] = t["trsv_512_lower"] + t["trsv_512_upper"]
put("mgs_block_4_4_512", "gmres", "k_mgs_block<double, 4, 4")
put("trsv_shell_rcm_lower", "shell_rcm", "k_trsv_sf<double, 0, false")
put("trsv_shell_rcm_upper", "shell_rcm", "k_trsv_sf<double, 1, true")
if "trsv_shell_rcm_lower" in fresh and "trsv_shell_rcm_upper" in fresh:
    t["trsv_shell_rcm"] = (t["trsv_shell_rcm_lower"] + t["trsv_shell_rcm_upper"]) // 2
put("spmv_csr_shell_rcm", "shell_rcm", "k_csr_wp<double")
put("mcsgs_512", "bicgstab_rb", "k_mc_rb<double>")
# the reference's own operator: 27-point Laplacian, 256^3
N = 256
n27 = N ** 3
nnz27 = (3 * N - 2) ** 3
t["spmv_csr_lap27_256_algorithmic"] = 12 * nnz27 + 4 * (n27 + 1) + 16 * n27
# (k_csr_tr<T, MODE, DOT, PAT, ...>: PAT = columns from the row-pattern dictionary, the stored columns not read)
# (k_csr_wr<T, MODE, DOT, PAT, GW>: the wave-private row walk rows of 16+ entries take since the end of round 6)
put("spmv_csr_lap27_256", "lap27_cg", "k_csr_wr<double, 0, true, true")
put("spmv_csr_lap27_256_columns_read", "lap27_cg", "k_csr_wr<double, 0, true, false")
put("spmv_ell_lap27_256", "lap27_ell", "k_ell2<double, 0, true, true, true")
put("spmv_ell_lap27_256_columns_read", "lap27_ell", "k_ell2<double, 0, true, true, false")
t["spmv_ell_lap27_256_algorithmic"] = 12 * 27 * n27 + 16 * n27
put("trsv_lap27_256_lower", "lap27_gmres", "k_trsv_box<double, true, true")
put("trsv_lap27_256_upper", "lap27_gmres", "k_trsv_box<double, false, false")
if "trsv_lap27_256_lower" in fresh and "trsv_lap27_256_upper" in fresh:
    t["trsv_lap27_256"] = (t["trsv_lap27_256_lower"] + t["trsv_lap27_256_upper"]) // 2
t["trsv_lap27_256_algorithmic"] = 12 * ((nnz27 - n27) // 2) + 8 * n27 + 4 * n27 + 16 * n27
# MC-SGS on the 27-point operator: 8 colours = 14 sweeps per apply (first forward, 6 forward, 6 backward, last backward); the
# number of applies in the pass = launches of the first forward sweep
sw = {k: v for k, v in raw.get("lap27_bicgstab", {}).items() if "k_mc_sweep<" in k and "FETCH_SIZE" in v and "WRITE_SIZE" in v}
if sw:
    applies = min(v["FETCH_SIZE"][1] for v in sw.values())
    t["mcsgs_lap27_256"] = int(round(sum((2 * v["FETCH_SIZE"][0] + v["WRITE_SIZE"][0]) * 1024 * v["FETCH_SIZE"][1] for v in sw.values()) / applies))
    t["mcsgs_lap27_256_sweeps_per_apply"] = int(round(sum(v["FETCH_SIZE"][1] for v in sw.values()) / applies))
    fresh.append("mcsgs_lap27_256")
t["mcsgs_lap27_256_algorithmic"] = 12 * (nnz27 - n27) + 40 * n27
t["fresh_in_round_6"] = fresh
t["notes"] = ("keys listed in fresh_in_round_6 come from this round's counter passes (tools/profile_r06.sh); the others are round 5's "
              "figures of kernels this round left alone.  trsv_lap27_256_*: the sheared-pencil solve of the 27-point stencil (k_trsv_box): "
              "packed coefficients without column indices + the outflow records.")
json.dump(t, open(os.path.join(here, "r06_traffic.json"), "w"), indent=1)
for k, v in t.items():
    if isinstance(v, int):
        print("%-40s %8.3f GB%s" % (k, v / 1e9, "   (round 6)" if k in fresh else ""))
