R=${GRAFT_REPO_ROOT:-$PWD}
cd $R
python -c "
import sys; sys.path.insert(0,'.')
from rocalution_amd import generators as gen
for k in ('rcm','random'): gen.shell_variant(549,k)
" 
for k in rcm delaunay random; do timeout 600 python tools/sf_soak.py $k 549 300; echo "rc=$?"; done
echo "--- three processes side by side"
for k in rcm rcm random; do ( timeout 900 python tools/sf_soak.py $k 549 300; echo "rc=$?" ) & done; wait
echo "--- five processes side by side, small"
for i in 1 2 3 4 5; do ( timeout 900 python tools/sf_soak.py rcm 200 1000; echo "rc=$?" ) & done; wait
