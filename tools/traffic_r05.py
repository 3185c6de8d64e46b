"""profiles/r05_pmc_raw.json (per-kernel FETCH_SIZE / WRITE_SIZE averages of tools/profile_r05.sh) -> profiles/r05_traffic.json,
the table bench.py's `traffic_for` reads.  bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 (MI355X_MICROARCH.md, HBM section:
both counters in KiB, FETCH_SIZE counts 64 B per 128-B request on gfx950 -- calibrated again this round, r05_pmc_FETCH_SIZE_calib.txt)."""
import json
import os
import sys

here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "profiles")
raw = json.load(open(os.path.join(here, "r05_pmc_raw.json")))
# a later pass over some legs only (gpurun_out/prof_txt, merged back from the box): its legs replace / extend the committed ones
later = os.path.join(here, "..", "gpurun_out", "prof_txt", "r05_pmc_raw.json")
if os.path.exists(later):
    for leg, v in json.load(open(later)).items():
        if v:
            raw[leg] = v
    json.dump(raw, open(os.path.join(here, "r05_pmc_raw.json"), "w"), indent=1)


def find(leg, *needles):
    hits = [(k, v) for k, v in raw[leg].items() if all(s in k for s in needles)]
    if len(hits) != 1:
        sys.exit("%s %r: %d kernels match" % (leg, needles, len(hits)))
    v = hits[0][1]
    return int(round((2 * v["FETCH_SIZE"][0] + v["WRITE_SIZE"][0]) * 1024))


n, nnz = 512 ** 3, 7 * 512 ** 3 - 6 * 512 ** 2
t = {"how": __doc__.replace("\n", " ")}
t["spmv_csr_512"] = find("cg", "k_csr_pat2<double, 0, true")
t["spmv_csr_512_plain"] = find("cg", "k_csr_pat2<double, 0, false")
t["spmv_csr_512_columns_read"] = find("cg", "k_csr_tr<double, 0, true")
t["spmv_csr_512_algorithmic"] = 12 * nnz + 4 * (n + 1) + 16 * n
t["cg_update_512"] = find("cg", "k_cg_update<double")
t["cg_direction_512"] = find("cg", "k_cg_direction<double")
t["trsv_512_lower"] = find("gmres", "k_trsv_lat<double, true")
t["trsv_512_upper"] = find("gmres", "k_trsv_lat<double, false")
t["trsv_512"] = (t["trsv_512_lower"] + t["trsv_512_upper"]) // 2
t["lusolve_512"] = t["trsv_512_lower"] + t["trsv_512_upper"]
t["trsv_512_algorithmic"] = 8321499136
t["mgs_block_4_4_512"] = find("gmres", "k_mgs_block<double, 4, 4")
t["trsv_shell_lower"] = find("shell", "k_trsv_rec<double, 0, false")
t["trsv_shell_upper"] = find("shell", "k_trsv_rec<double, 1, true")
t["trsv_shell"] = (t["trsv_shell_lower"] + t["trsv_shell_upper"]) // 2
t["trsv_shell_algorithmic"] = 345952650
if "shell_rcm" in raw:
    t["trsv_shell_rcm_lower"] = find("shell_rcm", "k_trsv_sf<double, 0, false")
    t["trsv_shell_rcm_upper"] = find("shell_rcm", "k_trsv_sf<double, 1, true")
    t["trsv_shell_rcm"] = (t["trsv_shell_rcm_lower"] + t["trsv_shell_rcm_upper"]) // 2
t["spmv_csr_shell"] = find("shell", "k_csr_w4<double")
t["spmv_csr_shell_algorithmic"] = 661765200
t["mc_sweep_forward_512"] = find("bicgstab", "k_mc_sweep<double, true, false, true, true")
t["mc_sweep_back_last_512"] = find("bicgstab", "k_mc_sweep<double, false, true, false")
t["mc_sweep_back_first_512"] = find("bicgstab", "k_mc_sweep<double, true, false, false, false")
t["mcsgs_512_sweeps"] = t["mc_sweep_forward_512"] + t["mc_sweep_back_last_512"] + t["mc_sweep_back_first_512"]
t["mcsgs_512"] = find("bicgstab_rb", "k_mc_rb<double>")
t["mcsgs_512_algorithmic"] = 15013511168
t["spmv_ell_512"] = find("ell", "k_ell2<double, 0, true, true, true")
t["spmv_ell_512_plain"] = find("ell", "k_ell2<double, 0, true, false, true")
t["spmv_ell_512_columns_read"] = find("ell", "k_ell2<double, 0, true, true, false")
t["spmv_ell_512_algorithmic"] = 13421772800
t["spmv_csr_512_fp32"] = find("mixed", "k_csr_pat2<float, 0, true")
t["spmv_csr_512_fp32_algorithmic"] = 8 * nnz + 4 * (n + 1) + 8 * n
t["cg_update_512_fp32"] = find("mixed", "k_cg_update<float")
t["notes"] = ("trsv_512_*: the lattice form (k_trsv_lat): coefficients packed per pencil without column indices, the solution "
              "written once and read once by nobody but the face records -- 14.09 GB for both triangles = 0.85 x the 16.64 GB the "
              "CSR triangles hold (round 4: 23.75 GB = 1.43 x).  mcsgs_512: the one-pass red-black lattice form (k_mc_rb); mcsgs_512_sweeps: the three colour "
              "sweeps it replaces on such operators (RAMD_MC_RB=0; the bicgstab / ell / hyb legs of the first profiling pass of "
              "the round).  spmv_csr_512_fp32_columns_read (k_csr_tr<float>) is not in the "
              "profiled command: null in the bench line.")
json.dump(t, open(os.path.join(here, "r05_traffic.json"), "w"), indent=1)
for k, v in t.items():
    if isinstance(v, int):
        print("%-34s %8.3f GB" % (k, v / 1e9))
