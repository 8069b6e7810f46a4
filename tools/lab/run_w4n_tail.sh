cd $GRAFT_REPO_ROOT
for r in 1 2; do for b in gemm_lab gemm_lab_tail1 gemm_lab_tail2; do for sh in siglip_o siglip_fc2; do
  for v in w4p_brs_o1 w4n_brs; do LAB_SHAPE=$sh timeout 120 tools/lab/$b $v 5 | grep '^{' | python -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('$b', d['shape'], d['variant'], round(d['tflops']), d['vs_ref'])"; done; done; done; done
