# round 5: where the stream-K forms of 3 / 8 bits and 32-wide groups start to pay (rows below the 512 of wide_sk_pays), then the default bench line
mkdir -p gpurun_out/r05b38s
for b in 3 8; do
  timeout 300 python tools/wide_sk_ab.py --bits $b --gs 32 --ms 128,256,384 --act 0 2>&1 | grep -v amdgpu.ids >> gpurun_out/r05b38s/ab_small.log
done
timeout 200 python tools/wide_sk_ab.py --bits 4 --gs 32 --ms 128,256,384 --act 0 --shapes 4096x11008 2>&1 | grep -v amdgpu.ids >> gpurun_out/r05b38s/ab_small.log
cat gpurun_out/r05b38s/ab_small.log
timeout 280 python bench.py > gpurun_out/r05b38s/bench.json 2> gpurun_out/r05b38s/bench.err; echo bench rc=$?
