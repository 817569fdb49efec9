#!/bin/bash
# where the host time of a small-batch training step goes (cProfile, molhiv-shaped B = 32)
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out/host
timeout 600 python scripts/train_step_molhiv.py --batch 32 --steps 50 2>/dev/null | tail -1 | cut -c100-230
timeout 600 python -c "
import cProfile, pstats, sys, types, torch
sys.argv=['x']
sys.path.insert(0,'scripts')
import train_step_molhiv as t
args=types.SimpleNamespace(batch=32, steps=60, warmup=10, layers=5, d=300)
dev=torch.device('cuda',0)
pr=cProfile.Profile()
pr.enable()
r=t.run(args, dev)
pr.disable()
print(r['ms_per_step'])
st=pstats.Stats(pr); st.sort_stats('tottime'); st.print_stats(45)
" 2>&1 | grep -v amdgpu > gpurun_out/host/molhiv_b32.txt
head -75 gpurun_out/host/molhiv_b32.txt | cut -c1-170
