#!/bin/bash
# Does the placement by spacers (vmm_spacer) and by taking everything in 64 MiB handles (vmm_interleave) survive a
# device whose memory another process has just used and released?
S=tools/scratch/vmm_spacer
echo "== fresh box"
timeout -k 5 120 $S 1000000 64 40 136 232 || exit 1
timeout -k 5 120 $S 1000000 64 40 136 232 || exit 1
echo "== after a process that wrote 250 GB and exited"
timeout -k 5 120 python -c "
import torch
t = torch.empty(250 * 10**9, dtype=torch.uint8, device='cuda'); t.fill_(3); torch.cuda.synchronize(); print('dirtied', t.numel())
" || exit 1
timeout -k 5 120 $S 1000000 64 40 136 232 || exit 1
timeout -k 5 120 $S 1000000 64 40 136 232 || exit 1
echo "== everything as 64 MiB handles (round 2's way), after the same"
timeout -k 5 120 python -c "
import torch
t = torch.empty(250 * 10**9, dtype=torch.uint8, device='cuda'); t.fill_(3); torch.cuda.synchronize(); print('dirtied', t.numel())
" || exit 1
( time timeout -k 5 200 tools/scratch/vmm_interleave 64 ) 2>&1 | cut -c1-160 || exit 1
echo "== spacers again"
timeout -k 5 120 $S 1000000 64 40 136 232 || exit 1
sleep 20
echo "== spacers after 20 s of rest"
timeout -k 5 120 $S 1000000 64 40 136 232 || exit 1
