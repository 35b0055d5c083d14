#!/usr/bin/env python3
"""The tail split of the encoder GEMMs (csrc/gemm_corr.hip): one frozen fp16c encoder pass of the bench batch, ALONE on the device,
on an unmasked stream and on the 240-CU stream of the training schedule, with the products planned for {0 = single launch, 240, 256}
CUs; per-projection launch times from the library's own events.
    python tools/encoder_tail_split.py"""
import ctypes, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from ruart_amd import hip, synth
from ruart_amd.arguments import default_opt
from ruart_amd.bert import bert_encode, _Buffers

dev = torch.device("cuda:0")
opt = default_opt(vocab_size=20000, cuda=True, device=dev, max_od_num=36, batch_size=64)
tr, _ = bench.build_trainer(opt, synth.bert_config(), dev)
b = tr.ToCUDA(synth.synthetic_batch(opt, 64, seed=7, n_q=30, n_ocr=100, n_od=36))
packed = b[0]["_ruart_index"].packed
W = tr.network.Bert.weights
lib = hip.load()
streams = {"unmasked": torch.cuda.Stream(device=dev), "240 CUs": hip.cu_masked_stream(240, dev)}
N = 10
for sname, st in streams.items():
    for cus in (0, 240, 256):
        W.c_model.tail_cus = cus
        bf = _Buffers()
        with torch.cuda.stream(st):
            for _ in range(3):
                bert_encode(W, packed, bf)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(N):
                bert_encode(W, packed, bf)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t0) / N * 1e3
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
        print("%-9s plan %3d CUs: pass %6.2f ms | per launch: QKV %5.0f  AO %5.0f  FF1 %5.0f  FF2 %5.0f us" % (sname, cus, dt, per[0] / k, per[1] / k, per[2] / k, per[3] / k), flush=True)
tr.close(final=True)
