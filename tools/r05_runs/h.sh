cd /root/repo
timeout 900 python tools/slab_probe.py 64
for c in 0 1; do
RAMD_CSR_PAT=0 RAMD_CSR_COL2=$c timeout 600 python bench.py --solver mixed --steps 10 --warmup 2 --no-cpu-baseline --no-reference-gpu 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('col2=$c', d['value'],'outer it/s', 'inner spmv', d['roofline']['avg_ms'], d['roofline']['frac'])"
done
