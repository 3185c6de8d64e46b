cd /root/repo
for N in 512 256; do
RAMD_LAT_NODEP=1 TAG=nodep timeout 300 python tools/trsv_time.py poisson $N
for w in 2 4 7; do RAMD_LAT_WGS_PER_CU=$w TAG=wgs$w timeout 300 python tools/trsv_time.py poisson $N; done
for w in 4 7; do RAMD_LAT_NODEP=1 RAMD_LAT_WGS_PER_CU=$w TAG=nodep_wgs$w timeout 300 python tools/trsv_time.py poisson $N; done
done
