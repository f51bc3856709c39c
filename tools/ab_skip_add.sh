#!/bin/bash
# A/B: skip-tensor gradients added inside sr3d_gated_act_bwd_sum (default) vs by autograd (SR3D_FUSE_SKIP_GRAD_ADD=0); same box
for st in fp32 bf16; do
  for v in 1 0 1 0; do
    echo "== $st SR3D_FUSE_SKIP_GRAD_ADD=$v"
    SR3D_FUSE_SKIP_GRAD_ADD=$v python bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-secondary --storage $st 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(round(d['ms_per_step'],2), {k: round(v['ms_per_step'],2) for k,v in d['hbm_bound_kernels'].items()})"
  done
done
