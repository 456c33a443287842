"""Instruction counts of the relabelling pass by phase: bursts of 7-frame launches (a grid no other launch has) with the
kernel's ablation switches, to be run under  rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES -d DIR -o p --
and read back with  python tools/pass_phase_probe.py --read DIR   (mean per wave and burst, in launch order)."""
import ctypes as C, glob, os, sqlite3, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
VARIANTS = [(0, "full"), (8, "no replay"), (4, "nothing eligible"), (12, "nothing eligible, no replay"), (2, "staging only"),
            (3, "staging, no window rows"), (35, "prologue loads only")]
REPS, WARM, NB = 12, 4, 7
if len(sys.argv) > 2 and sys.argv[1] == "--read":
    db = glob.glob(os.path.join(sys.argv[2], "**", "*.db"), recursive=True)[0]
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select dispatch_id, kernel_name, counter_name, value, grid_size_z from counters_collection "
                       "where kernel_name like '%k_update_pass%' order by dispatch_id").fetchall()
    by = {}
    for did, name, cn, v, gz in rows:
        if gz == NB:
            by.setdefault(did, {"rgbd": "<true" in name})[cn] = v
    seq = [by[k] for k in sorted(by)]
    per = WARM + REPS
    for i, (rgbd, (dbg, label)) in enumerate([(r, v) for r in (0, 1) for v in VARIANTS]):
        burst = seq[i * per + WARM:(i + 1) * per]
        if not burst:
            break
        w = sum(b["SQ_WAVES"] for b in burst) / len(burst)
        print("%-4s %-30s VALU %7.1f  SALU %7.1f per wave (%d waves, %d launches)" % (
            "rgbd" if rgbd else "rgb", label, sum(b["SQ_INSTS_VALU"] for b in burst) / len(burst) / w,
            sum(b["SQ_INSTS_SALU"] for b in burst) / len(burst) / w, w, len(burst)))
    sys.exit(0)
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import util
from supersurfel_fusion_amd import binding
lib = binding.load_lab()          # (the probe entry points live in the lab build: -DSSF_EXPERIMENTS)
lib.lib.ssf_dbg_time_pass.restype = C.c_double
lib.lib.ssf_dbg_time_pass.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
f = binding.Fusion(lib, util.make_cfg(lib, 640, 480, nb_supersurfels_max=50000, extract_batch=8))
for k in range(8):
    f.submit_frame(*util.frame(k, 640, 480))
f.process_submitted()
for rgbd in (0, 1):
    for dbg, label in VARIANTS:
        print(rgbd, label, "%.1f us" % lib.lib.ssf_dbg_time_pass(f.h, REPS, rgbd, dbg, NB))
