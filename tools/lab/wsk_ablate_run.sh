# on the GPU box: the stream-K kernel with parts of its K loop removed (timing only), 4096^2 and 4096x11008, M = 2048, fp16, no act-order
mkdir -p gpurun_out/r05abl
for n in 0 1 3 4 8 15; do
  if [ $n = 0 ]; then unset GPTQ_MI355X_LIB; else export GPTQ_MI355X_LIB=$PWD/tools/lab/libgptq_wsk_abl$n.so; fi
  echo "## GPTQ_WSK_ABL=$n" >> gpurun_out/r05abl/ablate.log
  timeout 120 python tools/wide_sk_ab.py --ms 2048 --act 0 --shapes 4096x4096,4096x11008 2>&1 | grep -v amdgpu.ids >> gpurun_out/r05abl/ablate.log
done
cat gpurun_out/r05abl/ablate.log
