#!/usr/bin/env python3
"""Phase times inside attn_flash_split_kernel (diagnostic build: tools/build_variant_bk.sh stamps -DRUART_ABL_ATTN_STAMPS, run with
RUART_HIP_LIB=build/libruart_hip_stamps.so): per workgroup s_memrealtime at entry, after the block descriptors, when the K / V / Q
loads have landed, after the LDS images + barrier, after the products, after the stores have drained; HW_ID gives the CU."""
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
hip.check(lib.ruart_bert_attention_split_set_heads(0), "heads")
nwg = nb * NH
st = torch.zeros(nwg * 8, dtype=torch.int64, device=dev)
def call():
    hip.check(lib.ruart_bert_attention_split(hip.ptr(qkv), 3 * H, hip.ptr(ctx16), hip.ptr(ctx8), H, H, NH, nb, hip.ptr(p.blk[0]), hip.ptr(p.blk[1]),
                                             hip.ptr(p.blk[2]), hip.ptr(p.blk[3]), hip.ptr(p.tok_lo), hip.ptr(p.tok_hi), None, hip.stream_ptr()), "attn")
for _ in range(3):
    call()
flush.fill_(1.0); torch.cuda.synchronize()
lib.ruart_attn_set_stamps(st.data_ptr()); call(); torch.cuda.synchronize(); lib.ruart_attn_set_stamps(None)
t = st.cpu().numpy().reshape(nwg, 8)
ts = t[:, :6].astype(np.float64) * 0.01
t0 = ts[:, 0].min()
hw = t[:, 6]; xcc = t[:, 7] & 0xf
cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7 if True else 0
cuid = xcc * 1000 + se * 100 + sh * 20 + cu
names = ["descriptors", "K/V/Q loads land", "split + LDS + barrier", "products", "stores drain"]
print("workgroups %d, kernel %.1f us" % (nwg, ts[:, 5].max() - t0))
for i, n in enumerate(names):
    d = ts[:, i + 1] - ts[:, i]
    print("  %-22s p10 %6.2f  p50 %6.2f  p90 %6.2f us" % (n, np.percentile(d, 10), np.median(d), np.percentile(d, 90)))
life = ts[:, 5] - ts[:, 0]
print("  %-22s p10 %6.2f  p50 %6.2f  p90 %6.2f us" % ("workgroup lifetime", np.percentile(life, 10), np.median(life), np.percentile(life, 90)))
ucu = np.unique(cuid)
print("distinct CUs seen %d; workgroups per CU: min %d max %d" % (len(ucu), min((cuid == c).sum() for c in ucu), max((cuid == c).sum() for c in ucu)))
# concurrency on one CU: at the entry of each workgroup, how many others of the same CU are alive
conc = []
for c in ucu[:32]:
    m = cuid == c
    s_, e_ = ts[m, 0], ts[m, 5]
    conc += [int(((s_ <= x) & (e_ > x)).sum()) for x in s_]
print("workgroups alive on a CU at a workgroup's entry (itself included): p10 %d p50 %d p90 %d max %d" % (np.percentile(conc, 10), np.median(conc), np.percentile(conc, 90), max(conc)))
# gap between an end on a CU and the next start on that CU
gaps = []
for c in ucu[:32]:
    m = cuid == c
    s_, e_ = np.sort(ts[m, 0]), np.sort(ts[m, 5])
    k = len(s_)
    occ = int(np.median(conc))
    if k > occ:
        gaps += list(s_[occ:] - e_[:k - occ])
print("start(k) - end(k - occupancy) on a CU: p50 %.2f p90 %.2f us" % (np.median(gaps), np.percentile(gaps, 90)))
one = cuid == ucu[0]
o = np.argsort(ts[one, 0])
print("CU %d time line (us from kernel start): entry / loads landed / end" % ucu[0])
for r in ts[one][o][:14]:
    print("   %7.2f  %7.2f  %7.2f" % (r[0] - t0, r[2] - t0, r[5] - t0))
