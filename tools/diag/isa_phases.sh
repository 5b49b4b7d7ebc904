#!/bin/bash
# usage: isa_phases.sh file.hip kernel-mangled-prefix  -> scratch / barrier / vmem landmarks of one kernel's ISA
set -e
SRC=$1; K=$2; D=/tmp/isa_$$; mkdir -p $D; cd $D
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -I /root/repo/vq-vae_amd/csrc -I /root/repo/include -c $SRC -o x.o -save-temps 2>/dev/null
S=$(ls *gfx950.s)
awk -v k="$K" '$0 ~ "^"k {f=1} f{print} f && /^\.Lfunc_end/{exit}' $S > k.s
echo "lines: $(wc -l < k.s)  -> $D/k.s"
grep -n "scratch_\|s_barrier\|^.LBB\|global_load_lds\|global_load_dwordx4\|global_store\|s_cbranch" k.s | awk '{print $1, $2, $3, $4}'
