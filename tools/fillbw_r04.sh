#!/bin/bash
# round 4: the store ceiling of the rollout kernel's OWN geometry — persistent tiles, each writing its envs' contiguous adj / node / obs blocks once per step into
# slot (step % T) of [T, ...] storage, nontemporal or ordinary stores, 3 or 4 streaming waves, no compute — at the three sizes bench.py prices against:
#   98 MB  c2 / c3 tile (4 envs: adj 64 KB + node 25.6 KB + obs) x 1024 tiles, ONE slot (Infinity-Cache-resident: not an HBM figure)
#   2.5 GB the same into 26 slots (what the default bench launch writes)
#   6 GB   c4 tile (2 envs x (32 x 72 x 72 adj + 32 x 72 x 8 node) floats) x 4096 tiles, one slot and 26 slots
# -> gpurun_out/r04_fillbw.json (copied to profiles/ by hand)
set -e
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 tools/tilebw.hip -o tools/tilebw.bin
{
./tools/tilebw.bin c2_one_slot 1024 4000 1600 132 1 520
./tools/tilebw.bin c2_26_slots 1024 4000 1600 132 26 520
./tools/tilebw.bin c4_one_slot 4096 82944 9216 208 1 40
./tools/tilebw.bin c4_26_slots 4096 82944 9216 208 26 52
} 2> gpurun_out/r04_fillbw.log | python3 -c "
import json, sys
rows = [json.loads(l) for l in sys.stdin if l.startswith('{')]
by = {r['geometry']: r for r in rows}
out = {'what': 'store ceiling of the rollout kernel\'s own store geometry (tools/tilebw.hip: persistent tiles, per-tile contiguous adj / node / obs blocks, slot-per-step storage, best of 3 / 4 streaming waves x plain / nontemporal stores, no compute), one MI355X',
       'fill_GBps': {'98MB': by['c2_one_slot']['best_GBps'], '2.5GB': by['c2_26_slots']['best_GBps'], '6GB': by['c4_one_slot']['best_GBps'], '158GB': by['c4_26_slots']['best_GBps']},
       'note': '98MB is Infinity-Cache-resident (not an HBM rate); the others are DRAM writes', 'geometries': rows, 'log': 'profiles/r04_fillbw.log'}
json.dump(out, open('gpurun_out/r04_fillbw.json', 'w'), indent=1); print(json.dumps(out['fill_GBps']))
"
