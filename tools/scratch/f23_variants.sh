#!/bin/bash
cd stylegan3-editing_amd/csrc
for v in "-DSG3_F23_ONE_TILE" ; do
  touch sg3_modconv_f23.hip; make EXTRA="$v" > /tmp/mk.log 2>&1 || tail -3 /tmp/mk.log
  echo "== variant [$v]"
  (cd ../.. && timeout -k 10 300 python tools/scratch/f23_repro.py 6 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-150)
done
