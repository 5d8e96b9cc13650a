#!/usr/bin/env python3
"""Times the REFERENCE's own CPU QuantLinear.forward (auto_gptq/nn_modules/qlinear/qlinear_cuda_old.py, loaded by file path:
`import auto_gptq` fails under transformers 5.x) next to this repo's oracle port, in the build container (the reference tree does
not travel to the GPU box).  BASELINE.md records the output.  Usage: python tools/time_reference_cpu.py [--threads N]"""
import argparse, importlib.util, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import gptq_oracle as O

REF = os.environ.get("GPTQ_REFERENCE", "/root/reference")


def load_ref_class(fname):
    path = os.path.join(REF, "auto_gptq/nn_modules/qlinear", fname)
    spec = importlib.util.spec_from_file_location("ref_" + fname[:-3], path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.QuantLinear


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--reps", type=int, default=5)
    args = ap.parse_args()
    if args.threads:
        torch.set_num_threads(args.threads)
    QL = load_ref_class("qlinear_cuda_old.py")
    print(f"torch {torch.__version__}, threads {torch.get_num_threads()}, reference class {QL.__module__}.QuantLinear (QUANT_TYPE={QL.QUANT_TYPE})")
    for K, N in ((4096, 4096), (4096, 11008), (11008, 4096)):
        L = O.random_quant_layer(K, N, 4, 128, seed=1)
        q = QL(4, 128, K, N, False)
        q.qweight, q.qzeros, q.scales, q.g_idx = L["qweight"], L["qzeros"], L["scales"], L["g_idx"]
        x = (torch.rand(1, K) - 0.5).half()
        with torch.no_grad():
            y_ref = q(x)
            y_or = O.forward(x, L["qweight"], L["qzeros"], L["scales"], L["g_idx"], None, 4, O.ZERO_WRAP)
            same = bool(torch.equal(y_ref, y_or))
            ts = []
            for fn in (lambda: q(x), lambda: O.forward(x, L["qweight"], L["qzeros"], L["scales"], L["g_idx"], None, 4, O.ZERO_WRAP)):
                fn()
                t0 = time.perf_counter()
                for _ in range(args.reps):
                    fn()
                ts.append((time.perf_counter() - t0) / args.reps)
        ab = K * N // 2 + (K // 128) * N // 2 + (K // 128) * N * 2 + 2 * K + 2 * N
        print(f"{K}x{N} M=1 fp16: reference class {ts[0] * 1e3:8.1f} ms ({ab / ts[0] / 1e9:.3f} GB/s) | oracle port {ts[1] * 1e3:8.1f} ms ({ab / ts[1] / 1e9:.3f} GB/s) | outputs bit-identical: {same}")


if __name__ == "__main__":
    main()
