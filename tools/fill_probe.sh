#!/bin/bash
# Where a 20-frame timed region goes: the driver's form of bench.py several times (pinned / not pinned to the GPU's NUMA
# node), each line's "pipeline_fill" key, then tools/startup_probe.py on the same box.
#   bash tools/gpu_run.sh <outdir> sh:"bash tools/fill_probe.sh"
cd ${GRAFT_REPO_ROOT:-.}
show() { tail -n 1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); p=d.get('pipeline_fill') or {}
t=p.get('frame_done_us') or []
print('$1', round(d['value']), 'frames/s; region', p.get('region_us'), 'us, call returned', p.get('call_returned_us'), 'us; frame 0 at', t[0] if t else None, 'last at', t[-1] if t else None)
print('   increments:', ' '.join('%d' % (b-a) for a,b in zip([0]+t[:-1], t)))
if 'track_entered_us' in p:
    e, r1, ic = p['track_entered_us'], p['first_icp_record_us'], p['icp_done_us']
    print('   entry->first record:', ' '.join('%d' % (b-a) for a,b in zip(e, r1)))
    print('   first record->icp done:', ' '.join('%d' % (b-a) for a,b in zip(r1, ic)))
    print('   icp done->frame done:', ' '.join('%d' % (b-a) for a,b in zip(ic, t)))
    print('   iterations:', p['icp_iters'], ' batches launched (us, frames):', p['extract_batches_launched'])"; }
for r in 1 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 --extras 0 --cpu-frames 0 --profile-frames 0 2>/dev/null | show pinned_$r; done
for r in 1 2; do python bench.py --gpus 1 --steps 20 --warmup 5 --extras 0 --cpu-frames 0 --profile-frames 0 --pin 0 2>/dev/null | show unpinned_$r; done
for e in "$@"; do
  for r in 1 2; do env $e python bench.py --gpus 1 --steps 20 --warmup 5 --extras 0 --cpu-frames 0 --profile-frames 0 2>/dev/null | show ${e}_$r; done
done
python tools/startup_probe.py 2>/dev/null
