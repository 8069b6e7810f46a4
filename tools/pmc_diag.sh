cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out
try() { name=$1; shift; rm -rf /tmp/pd_$name; timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d /tmp/pd_$name -o f -- "$@" > $O/pd_$name.out 2> $O/pd_$name.err; rc=$?; n=$(find /tmp/pd_$name -name '*counter_collection.csv' -exec wc -l {} \; 2>/dev/null | head -1); echo "pmcdiag $name rc=$rc rows=$n"; }
rocprofv3 -L 2>/dev/null | grep -i -E "FETCH_SIZE|WRITE_SIZE|TCC_EA0_RDREQ|TCC_EA0_WRREQ" | head -12 > $O/pd_counters.txt; head -12 $O/pd_counters.txt
try tiny python -c "import torch; x = torch.randn(1 << 20, device='cuda'); print(float(x.sum()))"
try big python -c "import torch; x = torch.randn((256000, 3584), device='cuda'); y = x.to(torch.bfloat16); print(float(y.float().sum()))"
try gen python -c "
import torch
g = torch.Generator(device='cuda'); g.manual_seed(3)
for i in range(300):
    x = torch.randn((3584, 3584), generator=g, device='cuda', dtype=torch.float32) * 0.02
print(float(x.sum()))"
try weights python -c "
import sys; sys.path.insert(0, '$R')
import torch
from vidi_amd import config as C
from vidi_amd.weights import init_random_weights
w = init_random_weights(C.vidi15_9b(), seed=3, dtype=torch.bfloat16, device='cuda:0'); torch.cuda.synchronize(); print(len(w))"
try b300 python $R/bench.py --frames 300 --steps 1 --warmup 0 --decode-steps 2 --no-cpu-baseline --no-kernel-timer --no-preproc --no-verify
