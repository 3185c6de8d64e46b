cd /root/repo
for N in 512 256 128; do
for w in 1 2 3 4 5; do RAMD_LAT_WGS_PER_CU=$w TAG=wgs$w timeout 300 python tools/trsv_time.py poisson $N; done
done
for w in 1 2 3 4 7; do RAMD_LAT_WGS_PER_CU=$w RAMD_SLAB_ONLY=gmres timeout 300 python tools/slab_probe.py 64; done
