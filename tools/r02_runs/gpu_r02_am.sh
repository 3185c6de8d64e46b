#!/bin/bash
mkdir -p gpurun_out/r02am
cd /root/repo
export TMPDIR=/tmp
python - <<'PY'
import re,sys
s=open('tests/test_gpu_box_tiles_forced.py').read()
ns={}
exec(s[s.index('_RAGGED = r"""'):s.index('@pytest.mark.gpu\n@pytest.mark.parametrize("dedup", ["0", "1"])\ndef test_ragged')], ns)
open('/tmp/ragged.py','w').write(ns['_RAGGED'] % {"root": "/root/repo"})
PY
RAMD_TRSV_CT_MINROWS=0 RAMD_TRSV_CT_MINLEN=0 RAMD_TRSV_CT_DEDUP=1 RAMD_TRSV_CT_VERBOSE=1 python /tmp/ragged.py 2>&1 | grep -v "^box-tile plan: box\|references" | cut -c1-230
