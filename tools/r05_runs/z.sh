# the counter / trace passes of the one-pass red-black MC-SGS legs (k_mc_rb) that the restored tree lacks
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out/r05z
mkdir -p $O
cd $R
bash tools/profile_r05.sh bicgstab_rb ell_rb hyb_rb > $O/profile.log 2>&1; tail -3 $O/profile.log
