# softmax attention ('embedded'): the new per-frame-group split-fp16 form vs the fused exact-fp32 kernels, same shapes
R=$GRAFT_REPO_ROOT
cd $R
{ echo "## per-frame-group contractions on the split-fp16 kernels (profiles/ubench/attn_chunked_probe.py)"; timeout -k 10 300 python profiles/ubench/attn_chunked_probe.py 5 2>&1 | grep -v amdgpu.ids;
  echo; echo "## fused exact-fp32 kernels (profiles/ubench/attn_softmax_only.py), same box"; timeout -k 10 300 python profiles/ubench/attn_softmax_only.py 3 2>&1 | grep -v amdgpu.ids; } > gpurun_out/r04_attn_softmax_chunked_vs_fused.txt
cat gpurun_out/r04_attn_softmax_chunked_vs_fused.txt
