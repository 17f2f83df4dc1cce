#!/bin/bash
# Register / LDS / scratch use of the kernels of one translation unit:  tools/kres.sh sweepga_amd/csrc/swg_segsort.hip [filter]
# (compiles the file with --save-temps in a scratch directory and reads the .amdhsa_ directives of the device assembly)
set -e
src=$(readlink -f "$1"); pat=${2:-.}
d=$(mktemp -d); cd "$d"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$src" -o x.o --save-temps >/dev/null 2>&1
grep -h "amdhsa_kernel \|amdhsa_next_free_vgpr\|amdhsa_group_segment_fixed\|amdhsa_private_segment_fixed" ./*gfx950.s | paste - - - - |
  sed 's/\t\+/ /g;s/ \+/ /g;s/\.amdhsa_kernel //;s/\.amdhsa_group_segment_fixed_size/lds/;s/\.amdhsa_private_segment_fixed_size/scratch/;s/\.amdhsa_next_free_vgpr/vgpr/' | grep -E "$pat" | cut -c1-260
cd /; rm -rf "$d"
