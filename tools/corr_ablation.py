#!/usr/bin/env python3
"""Which fp8 correction products of the fp16c encoder are needed to hold 1e-3?  

For every projection site (QKV, attention output, intermediate, output dense) the fp16c GEMM can carry both correction products
(3), only a_lo . w_hi - the activation's rounding residual - (1), only a_hi . w_lo - the weight's - (2), or none (0)
(ruart_gemm_16c_nt_sel / ruart_bert_set_correction).  This sweeps all 4^4 site settings - plus, for the full setting, switching
the correction off in one layer at a time - over the reference-generated goldens of tests/golden and prints, per setting,
max |p - p_ref| on each golden and the modelled GEMM cost relative to the full setting.

    python tools/corr_ablation.py [--quick] > profiles/r03_corr_ablation.log

Cost model: a site's product costs (1 + 0.39 * halves) * N * K (the fp8 phase with both halves takes 0.78 of the f16 phase's time,
profiles/r02_gemm_corr_bench.log: 1.78x); per layer QKV 2304*768, AO 768*768, FF1 3072*768, FF2 768*3072.
"""
import argparse
import itertools
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ruart_amd import hip, synth                                   # noqa: E402
from ruart_amd.arguments import default_opt                         # noqa: E402
import ruart_amd.layers as L                                        # noqa: E402

DEV = "cuda:0"
GOLD = os.path.join(ROOT, "tests", "golden")
SITE_W = np.array([2304 * 768, 768 * 768, 3072 * 768, 768 * 3072], dtype=np.float64)


def T(a):
    return torch.from_numpy(np.ascontiguousarray(a))


def build(z, cfg, **extra):
    from ruart_amd.sdnet import SDNet
    opt = default_opt(vocab_size=int(z["vocab_size"]), cuda=True, device=DEV, bert_precision="fp16c", bert_ln_fold=0, **extra)      # (the per-site knobs exist in the unfolded pass only)
    opt["bert_state"], opt["bert_config"] = synth.make_bert_weights(cfg, seed=int(z["seed"]), w_std=float(z["w_std"])), cfg
    sw = synth.make_sdnet_weights(opt, seed=int(z["seed"]))
    net = SDNet(opt, {"glove_embedding": T(sw["glove_embed.weight"]), "fast_embedding": T(sw["fast_embed.weight"])})
    net.load_state_dict({k: T(v) for k, v in sw.items()}, strict=False)
    return net.to(DEV), opt


def cases(quick):
    out = []
    z = np.load(os.path.join(GOLD, "sdnet_e2e_full.npz"))
    cfg = synth.bert_config(vocab_size=int(z["bert_vocab"]))
    net, opt = build(z, cfg, max_od_num=36)
    b = synth.synthetic_batch(opt, int(z["B"]), seed=int(z["batch_seed"]), n_q=30, n_ocr=100, n_od=36, bert_vocab=int(z["bert_vocab"]), ragged=bool(z["ragged"]))
    assert b[1]["num_cnt"] == z["ocr_num_cnt"].tolist()
    out.append(("bench B=64", net, b, z["scores"]))
    z = np.load(os.path.join(GOLD, "sdnet_e2e_full_ragged.npz"))
    cfg = synth.bert_config(vocab_size=int(z["bert_vocab"]))
    net, opt = build(z, cfg, max_od_num=36)
    b = synth.synthetic_batch(opt, int(z["B"]), seed=int(z["batch_seed"]), n_q=30, n_ocr=100, n_od=36, bert_vocab=int(z["bert_vocab"]), ragged=True)
    assert b[1]["num_cnt"] == z["ocr_num_cnt"].tolist()
    out.append(("ragged B=64 w0.05", net, b, z["scores"]))
    if not quick:
        z = np.load(os.path.join(GOLD, "sdnet_e2e_stress.npz"))
        cfgl = synth.bert_config(vocab_size=int(z["bert_vocab"]), hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096)
        net, opt = build(z, cfgl, BERT_LARGE=True, max_ocr_num=300, max_od_num=100, BERT_large_model_file="unused")
        b = synth.synthetic_batch(opt, int(z["B"]), seed=int(z["batch_seed"]), n_q=30, n_ocr=300, n_od=100, bert_vocab=int(z["bert_vocab"]), ragged=True)
        assert b[1]["num_cnt"] == z["ocr_num_cnt"].tolist()
        out.append(("stress bert-large B=2", net, b, z["scores"]))
    return out


def err(net, batch, ref):
    q, ocr, od, _, _ = batch
    L.set_dropout_prob(0.0)
    net.eval()
    net.drop_emb = False
    for d in (q, ocr, od):
        d.pop("_ruart_index", None)
    with torch.no_grad():
        s, _ = net(q, ocr, od)
    return float(np.abs(s.float().cpu().numpy() - ref).max())


def cost(sites, layer_frac=1.0):
    halves = np.array([{0: 0, 1: 1, 2: 1, 3: 2}[s] for s in sites], dtype=np.float64)
    full = (SITE_W * (1 + 0.39 * 2)).sum()
    return float(((SITE_W * (1 + 0.39 * halves * layer_frac)).sum()) / full)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--quick", action="store_true")
    a = ap.parse_args()
    lib = hip.load()
    cs = cases(a.quick)
    names = [c[0] for c in cs]
    print("golden cases: %s;  settings: site code per (QKV, AO, FF1, FF2): 3 both, 1 a_lo.w_hi, 2 a_hi.w_lo, 0 none" % ", ".join(names))
    rows = []
    for sites in itertools.product((3, 1, 2, 0), repeat=4):
        hip.check(lib.ruart_bert_set_correction(*sites, 0xFFFFFFFFFFFFFFFF), "set_correction")
        e = [err(net, b, ref) for (_, net, b, ref) in cs]
        rows.append((cost(sites), sites, e))
    rows.sort(key=lambda r: (r[0], max(r[2])))
    print("\n%-16s %-8s  %s   holds 1e-3 / with 2x margin" % ("sites", "cost", "  ".join("%-22s" % n for n in names)))
    for c, sites, e in rows:
        print("%-16s %-8.3f  %s   %s / %s" % (str(sites), c, "  ".join("%-22.2e" % x for x in e), max(e) < 1e-3, max(e) < 5e-4))
    ok = [r for r in rows if max(r[2]) < 5e-4]
    print("\ncheapest setting holding 1e-3 with 2x margin on all cases: %s cost %.3f" % (ok[0][1], ok[0][0]) if ok else "\nno setting holds the margin")
    # one layer at a time without correction (full sites elsewhere)
    print("\nfull setting with the correction switched off in ONE layer (bert-base cases):")
    for l in range(12):
        hip.check(lib.ruart_bert_set_correction(3, 3, 3, 3, 0xFFFFFFFFFFFFFFFF & ~(1 << l)), "set_correction")
        e = [err(net, b, ref) for (_, net, b, ref) in cs[:2]]
        print("  layer %2d off: %s" % (l, "  ".join("%.2e" % x for x in e)))
    print("\ncorrection only in the first n layers:")
    for n in range(0, 13, 2):
        hip.check(lib.ruart_bert_set_correction(3, 3, 3, 3, (1 << n) - 1), "set_correction")
        e = [err(net, b, ref) for (_, net, b, ref) in cs[:2]]
        print("  first %2d layers: %s   cost %.3f" % (n, "  ".join("%.2e" % x for x in e), cost((3, 3, 3, 3), n / 12.0)))
    hip.check(lib.ruart_bert_set_correction(3, 3, 3, 3, 0xFFFFFFFFFFFFFFFF), "set_correction")
    for (_, net, _, _) in cs:
        net.Bert.close()


if __name__ == "__main__":
    main()
