#!/bin/bash
# timing prototype: the column attention's out_proj folded into the per-site context (-DPF_COLAPPLY_U=1; the values
# of U are a stand-in, so no parity run) against the same source with the MFMA column apply.  The prototype is
# 14_ucol_prototype.patch (plus `B * Lloc * 256` floats for ctx in pf_lib.hip::workspace_bytes); it was not kept:
#   patch -p0 < tools/runs/r06/14_ucol_prototype.patch; python tools/build_variant.py lib_ucol.so -DPF_COLAPPLY_U=1; python tools/build_variant.py lib_head.so
cd "$(dirname "$0")/../../.."
O=gpurun_out/r06y; mkdir -p $O
PF_AB_STEPS=10 python tools/flag_compare.py lib_head.so lib_ucol.so lib_head.so lib_ucol.so lib_head.so lib_ucol.so > $O/ab.txt 2>&1; cat $O/ab.txt
python tools/kernel_ab.py main lib_head.so lib_ucol.so > $O/main_ab_one_stream.txt 2>&1; cat $O/main_ab_one_stream.txt
