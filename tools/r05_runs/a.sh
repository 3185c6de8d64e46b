# first timings of the lattice form at 512^3 / 256^3 / the slab against the box tiles
cd /root/repo
for lat in 1 0; do
  for N in 512 256; do
    RAMD_TRSV_LAT=$lat TAG=lat$lat timeout 300 python tools/trsv_time.py poisson $N
  done
  RAMD_TRSV_LAT=$lat RAMD_SLAB_ONLY=gmres timeout 300 python tools/slab_probe.py 64
done
RAMD_TRSV_LAT=1 timeout 300 python bench.py --solver gmres --precond ilu0 --steps 60 --warmup 30 2>&1 | tail -3
