cd /root/repo
timeout 600 python tools/r05_runs/lat_check.py 2>&1 | grep -v "^grid\|^lattice plan" | tail -8
for N in 512 256; do
TAG=lat timeout 300 python tools/trsv_time.py poisson $N
RAMD_LAT_WGS_PER_CU=3 TAG=wgs3 timeout 300 python tools/trsv_time.py poisson $N
RAMD_LAT_NODEP=1 TAG=nodep timeout 300 python tools/trsv_time.py poisson $N
done
RAMD_SLAB_ONLY=gmres timeout 300 python tools/slab_probe.py 64
