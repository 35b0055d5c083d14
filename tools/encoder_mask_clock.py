#!/usr/bin/env python3
"""Is the frozen-encoder pass bound by CUs or by the chip's power / clock?  One fp16c pass of the bench batch, alone on the device, on
streams restricted to N CUs; per-projection launch times from the library's own events.  Under
`rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE -- python3 tools/encoder_mask_clock.py --cus N` the GEMM kernels' effective clock at
that width is GRBM_GUI_ACTIVE / 8 / duration (tools/pmc_kernel_table.py).
    python tools/encoder_mask_clock.py [--cus 128,192,224,240,256]"""
import argparse, ctypes, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ruart_amd import hip, synth
from ruart_amd.arguments import default_opt
from ruart_amd.bert import bert_encode, _Buffers

ap = argparse.ArgumentParser()
ap.add_argument("--cus", default="128,160,192,224,240,256")
ap.add_argument("--passes", type=int, default=10)
ap.add_argument("--batch", type=int, default=64, help="samples in the packed stream (128: what one pass over two batches would encode)")
a = ap.parse_args()
dev = torch.device("cuda:0")
opt = default_opt(vocab_size=20000, cuda=True, device=dev, max_od_num=36, batch_size=a.batch)
tr, _ = bench.build_trainer(opt, synth.bert_config(), dev)
b = tr.ToCUDA(synth.synthetic_batch(opt, a.batch, seed=7, n_q=30, n_ocr=100, n_od=36))
packed = b[0]["_ruart_index"].packed
W = tr.network.Bert.weights
W.c_model.tail_cus = 0
lib = hip.load()
for cus in [int(c) for c in a.cus.split(",")]:
    st = hip.cu_masked_stream(cus, dev) if cus < 256 else torch.cuda.Stream(device=dev)
    bf = _Buffers()
    with torch.cuda.stream(st):
        for _ in range(3):
            bert_encode(W, packed, bf)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.passes):
            bert_encode(W, packed, bf)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / a.passes * 1e3
        hip.check(lib.ruart_prof_enable(1), "prof")
        bert_encode(W, packed, bf)
        torch.cuda.synchronize()
        n = 64
        bt, et, fl = (ctypes.c_float * n)(), (ctypes.c_float * n)(), (ctypes.c_double * n)()
        cnt = ctypes.c_int()
        hip.check(lib.ruart_prof_timeline(bt, et, fl, n, ctypes.byref(cnt)), "timeline")
        lib.ruart_prof_enable(0)
    per = [0.0] * 4
    for i in range(cnt.value):
        per[i % 4] += (et[i] - bt[i]) * 1e3
    k = max(1, cnt.value // 4)
    tiles = [(packed.Tp // 256) * nn for nn in (9, 3, 12, 3)]
    rounds = " ".join("%.2f" % (t / cus) for t in tiles)
    print("B %d rows %d, %3d CUs: pass %6.2f ms | QKV %5.0f  AO %5.0f  FF1 %5.0f  FF2 %5.0f us | tiles / CUs: %s" % (a.batch, packed.Tp, cus, dt, per[0] / k, per[1] / k, per[2] / k, per[3] / k, rounds),
          flush=True)
    if cus < 256:
        hip.destroy_stream(st)          # a CU-masked queue alive at exit takes rocprofv3's teardown down (hip.destroy_stream)
tr.close(final=True)
