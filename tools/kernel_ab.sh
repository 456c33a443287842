#!/bin/bash
# per-kernel launch times (bench.py's hipEvent brackets, 8 frames per launch) of the product against variant builds, alternated:
#   bash tools/kernel_ab.sh <outdir> <rounds> <kernel,kernel,...> <variant|product|env:K=V|var:<variant>:K=V> ...
R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/${1:?outdir}; mkdir -p $O; cd $R
N=${2:?rounds}; K=${3:?kernels}; shift 3
for r in $(seq 1 $N); do for V in "$@"; do
  unset SSF_PRODUCT_VARIANT SSF_PASS_WAVES SSF_PASS_TEAM
  case $V in product) ;; env:*) export SSF_PRODUCT_VARIANT=lab; export "${V#env:}";; var:*) W=${V#var:}; export SSF_PRODUCT_VARIANT=${W%%:*}; export "${W#*:}";; *) export SSF_PRODUCT_VARIANT=$V;; esac
  timeout 300 python bench.py --extras 0 --cpu-frames 0 --steps 240 2>/dev/null | tail -n 1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['per_kernel']
print('$V run $r', round(d['value'],1), ' '.join('%s %.2f' % (n, k[n]['avg_us']) for n in '$K'.split(',') if n in k))" >> $O/kernel_ab.txt
done; done
cat $O/kernel_ab.txt
