#!/usr/bin/env python3
"""Step phases inside attn_flash_split_mh_kernel (diagnostic build -DRUART_ABL_ATTN_STAMPS): per workgroup and step (first four):
top of step, loads landed, LDS images written + next loads issued, barrier passed, products done, end of step."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ruart_amd import hip, synth
from ruart_amd.arguments import default_opt
from ruart_amd.batch import BatchIndex
dev = torch.device("cuda:0")
lib = hip.load()
lib.ruart_attn_set_stamps.argtypes = [ctypes.c_void_p]; lib.ruart_attn_set_stamps.restype = ctypes.c_int
opt = default_opt(vocab_size=20000, cuda=True, device=dev, bert_precision="fp16c", max_od_num=36, batch_size=64)
opt["bert_config"] = synth.bert_config()
q, ocr, od, gt, _ = synth.synthetic_batch(opt, 64, seed=7, n_q=30, n_ocr=100, n_od=36)
p = BatchIndex(q, ocr, od, opt, dev, pack=True, mfma_long=True).packed
H, NH = 768, 12
T, Tp, nb = p.T, p.Tp, p.n_blocks
qkv = (torch.randn(Tp, 3 * H) * 1.5).to(dev)
ctx16 = torch.zeros(Tp, H, dtype=torch.float16, device=dev); ctx8 = torch.zeros(Tp, 2 * H, dtype=torch.uint8, device=dev)
flush = torch.empty(128 << 20, dtype=torch.float32, device=dev)
HPG = int(sys.argv[1]) if len(sys.argv) > 1 else 4
hip.check(lib.ruart_bert_attention_split_set_heads(HPG), "heads")
nwg = nb * (NH // HPG)
st = torch.zeros(nwg * 32, dtype=torch.int64, device=dev)
def call():
    hip.check(lib.ruart_bert_attention_split(hip.ptr(qkv), 3 * H, hip.ptr(ctx16), hip.ptr(ctx8), H, H, NH, nb, hip.ptr(p.blk[0]), hip.ptr(p.blk[1]),
                                             hip.ptr(p.blk[2]), hip.ptr(p.blk[3]), hip.ptr(p.tok_lo), hip.ptr(p.tok_hi), None, hip.stream_ptr()), "attn")
for _ in range(3):
    call()
flush.fill_(1.0); torch.cuda.synchronize()
lib.ruart_attn_set_stamps(st.data_ptr()); call(); torch.cuda.synchronize(); lib.ruart_attn_set_stamps(None)
t = st.cpu().numpy().reshape(nwg, 32)
ts = t[:, :24].astype(np.float64).reshape(nwg, 4, 6) * 0.01
entry = t[:, 29].astype(np.float64) * 0.01
t0 = entry.min()
print("heads per workgroup %d, workgroups %d, last stamp at %.1f us" % (HPG, nwg, ts.max() - t0))
print("  entry -> top of step 0 (descriptors, first loads issued): p50 %.2f p90 %.2f us" % (np.median(ts[:, 0, 0] - entry), np.percentile(ts[:, 0, 0] - entry, 90)))
names = [("wait for the loads", 0, 1), ("split + LDS + stores + issue next + barrier", 1, 3), ("products + barrier", 3, 5)]
for j in range(min(4, HPG)):
    print("  step %d:" % j + "".join("  %s p50 %.2f p90 %.2f |" % (n, np.median(ts[:, j, b_] - ts[:, j, a_]), np.percentile(ts[:, j, b_] - ts[:, j, a_], 90)) for n, a_, b_ in names))
    print("          whole step p50 %.2f p90 %.2f us" % (np.median(ts[:, j, 5] - ts[:, j, 0]), np.percentile(ts[:, j, 5] - ts[:, j, 0], 90)))
hw = t[:, 30]; xcc = t[:, 31] & 0xf
cuid = xcc * 1000 + ((hw >> 13) & 7) * 100 + ((hw >> 12) & 1) * 20 + ((hw >> 8) & 0xf)
one = cuid == np.unique(cuid)[0]
o = np.argsort(entry[one])
print("one CU: entry / top of steps 0..3 (us from kernel start)")
for e, r in list(zip(entry[one][o], ts[one][o]))[:10]:
    print("   %7.2f | " % (e - t0) + "  ".join("%7.2f" % (r[j, 0] - t0) for j in range(min(4, HPG))))
