R=${GRAFT_REPO_ROOT:-$(pwd)}; O=$R/gpurun_out/r12c; mkdir -p $O; cd $R
run() { tag=$1; shift; timeout 300 python bench.py --extras 0 --cpu-frames 0 --steps 240 2>/dev/null | tail -n 1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['per_kernel']
print('$tag', round(d['value'],1), ' '.join('%s %.2f' % (n, k[n]['avg_us']) for n in ('passes_team_rgbd','passes_team_rgb','update_pass_rgbd','update_pass_rgb') if n in k))" >> $O/team_ab.txt; }
export SSF_PRODUCT_VARIANT=tnc
SSF_PASS_TEAM_WGS=128 run nocoh_team105
SSF_PASS_TEAM_WGS=64 run nocoh_team63
export SSF_PRODUCT_VARIANT=lab
SSF_PASS_TEAM_WGS=64 run team63
SSF_PASS_TEAM_WGS=40 run team40
cat $O/team_ab.txt
