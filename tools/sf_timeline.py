"""read the per-unit timestamps of k_trsv_sf (RAMD_TRSV_SF_DBG=<prefix>): hand-off and compute time along the dependency levels
    python tools/sf_timeline.py <prefix>_lower.bin"""
import sys
import numpy as np
f = open(sys.argv[1], "rb")
nunits, n, nwg, lpr = np.fromfile(f, np.int32, 4)
t = np.fromfile(f, np.uint64, 2 * nunits).reshape(nunits, 2).astype(np.int64)
ui = np.fromfile(f, np.int32, 4 * nunits).reshape(nunits, 4)
pu = np.fromfile(f, np.int32, n)
ck = np.fromfile(f, np.uint64, 4).astype(np.int64)
if len(ck) == 4 and ck[3] > ck[1]:
    print("shader clock during the kernel (block 0): %.0f MHz" % ((ck[2] - ck[0]) / ((ck[3] - ck[1]) * 0.01)))
cy = np.fromfile(f, np.uint64, 2 * nunits).reshape(nunits, 2)
if len(cy) == nunits:
    c1, c2, c3 = (cy[:, 0] & np.uint64(0xffffffff)).astype(np.int64), (cy[:, 0] >> np.uint64(32)).astype(np.int64), cy[:, 1].astype(np.int64)
    print("shader cycles from 'dependencies seen': chain done %.0f (median), result final %.0f, publication issued %.0f, after the second wall-clock read %s"
          % (np.median(c1), np.median(c2), np.median(c3), ""))
tick = 0.01  # wall_clock64: 100 MHz -> microseconds
tg, tp = t[:, 0] * tick, t[:, 1] * tick
t0 = tg.min()
tg, tp = tg - t0, tp - t0
ld = ui[:, 3]
has = ld >= 0
prod = np.where(has, pu[np.maximum(ld, 0)], 0)
hand = (tg - tp[prod])[has]          # from the publication of the last dependency to "all dependencies seen"
comp = tp - tg
print("units %d, waves %d, lanes per row %d, span %.1f us" % (nunits, nwg, lpr, tp.max()))
print("compute (dependencies seen -> published): mean %.2f  median %.2f  p90 %.2f us" % (comp.mean(), np.median(comp), np.percentile(comp, 90)))
for name, v in (("hand-off from the LAST dependency's publication", hand),):
    print("%s: mean %.2f median %.2f p10 %.2f p90 %.2f us; negative (dependency was not the last to arrive) %.1f %%"
          % (name, v.mean(), np.median(v), np.percentile(v, 10), np.percentile(v, 90), 100.0 * (v < 0).mean()))
# the critical path: walk back from the unit published last along the last-arriving dependency (approximated by the last dependency)
u = int(np.argmax(tp))
path = []
while True:
    path.append(u)
    if ld[u] < 0:
        break
    u = int(pu[ld[u]])
path = np.array(path[::-1])
print("path along last dependencies from the unit published last: %d units, %.1f us" % (len(path), tp[path[-1]] - tp[path[0]]))
ph = tg[path[1:]] - tp[path[:-1]]
pc = comp[path[1:]]
print("  on that path: hand-off mean %.2f median %.2f us (sum %.0f), compute mean %.2f us (sum %.0f)" % (ph.mean(), np.median(ph), ph.sum(), pc.mean(), pc.sum()))
