#!/bin/bash
for a in "--shape enc --bs 4 --dir fwd --variant 0" "--shape enc --bs 4 --dir bwd --variant 0" "--shape dec --bs 4 --dir both --variant 0" "--shape micro --bs 2 --dir both --variant 0"; do
  timeout -k 5 120 python tools/msda_probe.py $a --iters 30 2>&1 | grep -v amdgpu.ids | tail -2
done
