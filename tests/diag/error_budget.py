"""CPU emulation of the engine's rounding points on the REFERENCE's own full-size runs: which rounding is left in the sampled-logit error?

For a subset of the steps of a recorded run (tests/golden/sample_full*.npz, teacher-forced like maskbit_amd/parity_replay.py) the forward is evaluated in
fp32 on the CPU with each of the engine's rounding points switchable:

  x     LayerNorm outputs feeding QKV / FFN-up          att   attention outputs feeding out-proj        h    GELU outputs feeding FFN-down
  qkv   the packed q / k / v rows the attention reads    p     the probabilities of the PV product
  wt    trunk GEMM weights                               wh    the two head GEMMs' weights

each "f16" (rounded to fp16), "exact", and for the guided forward in the engine's DIFFERENTIAL form (the unconditional stream's operand is
fp16(x_c) + fp16(x_u - x_c)) or as two independent streams.  qkv additionally "diff" (q_u stored as fp16(q_c) + fp16(q_u - q_c)).  wt "corr4": fp16
weights + the engine's MX-fp4 correction (e2m1 of the operand values, block scales per `blk` columns, against e2m1 of W - fp16(W), one scale per weight
row) on the conditional rows.  Reported per case: rms of the centred error of the SAMPLED logits c + s (c - u) over the masked positions, and the token
mismatch against the reference's recorded predictions with the reference's noise.

usage: python tests/diag/error_budget.py <run name> [every-nth step] [case filter substring ...]
"""
import math
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from maskbit_amd import parity_replay as R, synth          # noqa: E402
from oracle import maskbit_oracle as O                       # noqa: E402

h16 = lambda t: t.to(torch.float16).to(torch.float32)
E2M1 = torch.tensor([0.0, 0.5, 1.0, 1.5, 2.0, 3.0, 4.0, 6.0])


def q_e2m1(v, blk, skip=0):
    """e2m1 with one power-of-two scale per `blk` consecutive columns (0 = per row), scaled so that 3 < max <= 6 (mb_common.h fp4_nosat_exp).
    skip = n (study): the scale is taken from the (n + 1)-th largest magnitude of the block and the n larger ones saturate at 6 scale units -- what a
    block with a massive-activation channel would need for the other 63 values to keep their resolution."""
    shp = v.shape
    w = v.reshape(-1, shp[-1]) if blk == 0 else v.reshape(-1, blk)
    am = (w.abs().amax(-1, keepdim=True) if skip == 0 else w.abs().topk(skip + 1, dim=-1).values[:, -1:]).clamp_min(1e-30)
    e = torch.floor(torch.log2(am))
    mant = am / torch.exp2(e)
    e = e + (mant > 1.5).float()                     # amax * 2^-(e-2) in (3, 6]
    sc = torch.exp2(e - 2)
    a = (w / sc).abs().clamp(max=6.0)
    idx = torch.bucketize(a, (E2M1[1:] + E2M1[:-1]) / 2)        # nearest grid point (ties: up; the hardware rounds to even -- immaterial here)
    return (torch.sign(w) * E2M1[idx] * sc).reshape(shp)


E2M3 = torch.tensor([i * 0.125 for i in range(8)] + [1.0 + i * 0.125 for i in range(8)] + [2.0 + i * 0.25 for i in range(8)] + [4.0 + i * 0.5 for i in range(8)])


def q_grid(v, blk, grid):
    """Block-scaled quantisation onto `grid` (non-negative magnitudes, sign separate): one power-of-two scale per `blk` columns (0 = per row), the
    largest magnitude of a block never saturates (scaled into (max / 2, max])."""
    shp = v.shape
    w = v.reshape(-1, shp[-1]) if blk == 0 else v.reshape(-1, blk)
    am = w.abs().amax(-1, keepdim=True).clamp_min(1e-30)
    gmax = float(grid[-1])
    e = torch.ceil(torch.log2(am / gmax))              # am / 2^e in (gmax / 2, gmax]
    sc = torch.exp2(e)
    a = (w / sc).abs().clamp(max=gmax)
    idx = torch.bucketize(a, (grid[1:] + grid[:-1]) / 2)
    return (torch.sign(w) * grid[idx] * sc).reshape(shp)


def q_e2m1_best(W):
    """e2m1 of weight VALUES with the per-row scale that minimises the row's quantisation error (mb::w4_from_f32): no-saturation scale or one binade lower."""
    a = q_e2m1(W, 0)
    b = q_e2m1(W * 2, 0) / 2                             # not a different scale by itself (q_e2m1 rescales) -- kept for symmetry
    w = W.reshape(-1, W.shape[-1])
    am = w.abs().amax(-1, keepdim=True).clamp_min(1e-30)
    e = torch.floor(torch.log2(am)); mant = am / torch.exp2(e); e = e + (mant > 1.5).float()
    best, berr = None, None
    for de in (0, -1):
        sc = torch.exp2(e - 2 + de)
        aa = (w / sc).abs().clamp(max=6.0)
        idx = torch.bucketize(aa, (E2M1[1:] + E2M1[:-1]) / 2)
        qq = torch.sign(w) * E2M1[idx] * sc
        err = (qq - w).pow(2).sum(-1, keepdim=True)
        if best is None: best, berr = qq, err
        else:
            take = err < berr
            best = torch.where(take, qq, best); berr = torch.where(take, err, berr)
    return best.reshape(W.shape)


class Emu:
    def __init__(self, sd, cfg, o):
        self.sd, self.cfg, self.o = sd, cfg, o
        self.w16 = {k: h16(v) for k, v in sd.items() if v.dim() == 2}
        self.wlo4 = {}
        self.w4 = {}

    def rnd(self, key, t):
        return h16(t) if self.o[key] == "f16" else t

    def weight(self, name, head=False):
        m = self.o["wh" if head else "wt"]
        if not head and self.o.get("wt_exact_for") and any(t in name for t in self.o["wt_exact_for"]):
            m = "exact"                                  # partial coverage: only these trunk GEMMs have their weight rounding corrected
        return self.sd[name] if m == "exact" else self.w16[name]

    def lin(self, name, bias, xc, xu, key, head=False):
        """One Linear over the conditional rows xc and (guided) the unconditional rows xu -> (out_c, out_u)."""
        W = self.weight(name, head)
        b = self.sd[bias]
        corr = (not head) and self.o["wt"] in ("corr4", "corr6")
        if head:
            ac = xc
        else:
            ac = self.rnd(key, xc)
        oc = F.linear(ac, W)
        if corr:
            if name not in self.wlo4:
                err = self.sd[name] - self.w16[name]
                if self.o["wt"] == "corr6":                                  # e2m3 weight-error operand (study)
                    self.wlo4[name] = q_grid(err, self.o.get("wblk", 0), E2M3)
                else:
                    q1 = q_e2m1(err, self.o.get("wblk", 0))
                    if self.o.get("wlevels", 1) == 2:                         # second e2m1 set on what the first one left (study)
                        q1 = q1 + q_e2m1(err - q1, self.o.get("wblk", 0))
                    self.wlo4[name] = q1
            blk = self.o.get("blk", 64)
            skip = self.o.get("skip", 0) if (key == "x" or not self.o.get("skip_x_only")) else 0
            ntop = self.o.get("tok_top", 0) if (key in self.o.get("tok_top_keys", ("x",))) else 0
            xs = xc
            if ntop:                                                          # study: the n largest channels leave the e2m1 token operand (their block scales
                sel = xc.abs().reshape(-1, xc.shape[-1]).mean(0).topk(ntop).indices   # then serve the other values) and meet the EXACT weight error in an fp16 K-tile
                xs = xc.clone(); xs[..., sel] = 0.0
                oc = oc + F.linear(h16(xc[..., sel]), (self.sd[name] - self.w16[name])[:, sel])
            xq = xs if blk < 0 else (q_grid(xs, blk, E2M3) if self.o.get("tok6") else q_e2m1(xs, blk, skip))
            oc = oc + F.linear(xq, self.wlo4[name])
        # activation-lo corrections on the conditional rows (study, round 6): what the fp16 rounding of this operand leaves, multiplied by ...
        alo = (self.o.get("alo") or {}).get(key) if not head else None
        if alo and self.o[key] == "f16":
            lo = xc - h16(xc)
            if alo == "f16":                                                  # ... the fp16 weight (a full second sweep)
                oc = oc + F.linear(h16(lo), W)
            elif alo == "e2m1":                                               # ... e2m1 of the weight values, lo as e2m1 with 64-column scales (a mini-tile set)
                if name not in self.w4:
                    self.w4[name] = q_e2m1_best(self.w16[name]) if not self.o.get("w4blk") else q_e2m1(self.w16[name], self.o["w4blk"])
                ntop = self.o.get("tok_top", 0) if (key in self.o.get("tok_top_keys", ("x",))) else 0
                if ntop:                                                      # the same channels' lo halves meet the fp16 weight in that K-tile
                    sel = xc.abs().reshape(-1, xc.shape[-1]).mean(0).topk(ntop).indices
                    los = lo.clone(); los[..., sel] = 0.0
                    oc = oc + F.linear(h16(lo[..., sel]), W[:, sel]) + F.linear(q_e2m1(los, 64), self.w4[name])
                else:
                    oc = oc + F.linear(q_e2m1(lo, 64), self.w4[name])
            elif alo == "e2m3":                                               # ... both operands e2m3 (FP6 at the FP4 rate)
                if name not in self.w4:
                    self.w4[name] = q_grid(self.w16[name], 0, E2M3)
                oc = oc + F.linear(q_grid(lo, 64, E2M3), self.w4[name])
            elif alo.startswith("top"):                                       # ... fp16 weights on the `n` channels with the largest mean |x| (outlier K-tile)
                n = int(alo[3:])
                sel = xc.abs().reshape(-1, xc.shape[-1]).mean(0).topk(n).indices
                oc = oc + F.linear(h16(lo[..., sel]), W[:, sel])
        if xu is None:
            return oc + b, None
        if head:
            return oc + b, F.linear(xu, W) + b
        if self.o["pair"]:
            ad = self.rnd(key, xu - xc)
            return oc + b, oc + F.linear(ad, W) + b
        ou = F.linear(self.rnd(key, xu), W)
        if corr:
            ou = ou + F.linear(q_e2m1(xu, self.o.get("blk", 64)), self.wlo4[name])
        return oc + b, ou + b

    def attention(self, qkv):
        cfg = self.cfg
        b = qkv.shape[0]
        d, H = cfg.hidden, cfg.heads
        dh = d // H
        qq, kk, vv = [t.reshape(b, -1, H, dh).transpose(1, 2) for t in qkv.split(d, -1)]
        s = (qq @ kk.transpose(-1, -2)) * (1 / math.sqrt(dh))
        p = torch.exp(s - s.amax(-1, keepdim=True))
        den = p.sum(-1, keepdim=True)
        return ((self.rnd("p", p) @ vv) / den).transpose(1, 2).reshape(b, -1, d)

    def forward(self, tokens, labels, guided):
        sd, cfg = self.sd, self.cfg
        B = tokens.shape[0]
        x_tok = F.linear(O.token_bit_vectors(tokens, cfg), sd["input_proj.weight"], sd["input_proj.bias"])

        def embed(lab):
            x = torch.cat([x_tok, sd["class_emb.weight"][lab].unsqueeze(1)], 1) + sd["pos_emb"]
            return O._ln(x, sd, "first_layer.0", 1e-12)
        xc = embed(labels)
        xu = embed(torch.full_like(labels, cfg.nclass)) if guided else None
        for l in range(cfg.depth):
            a, f = f"transformer.layers.{l}.0", f"transformer.layers.{l}.1"
            qc, qu = self.lin(a + ".mha.in_proj_weight", a + ".mha.in_proj_bias", xc, xu, "x")
            if self.o["qkv"] == "diff" and guided:
                qcr = h16(qc)
                qur = qcr + h16(qu - qc)
            else:
                qcr, qur = self.rnd("qkv", qc), (self.rnd("qkv", qu) if guided else None)
            ac = self.attention(qcr)
            au = self.attention(qur) if guided else None
            oc, ou = self.lin(a + ".mha.out_proj.weight", a + ".mha.out_proj.bias", ac, au, "att")
            xc = O._ln(oc + xc, sd, a + ".norm", 1e-12)
            if guided:
                xu = O._ln(ou + xu, sd, a + ".norm", 1e-12)
            uc, uu = self.lin(f + ".net.0.weight", f + ".net.0.bias", xc, xu, "x")
            hc = F.gelu(uc)
            hu = F.gelu(uu) if guided else None
            dc, du = self.lin(f + ".net.2.weight", f + ".net.2.bias", hc, hu, "h")
            xc = O._ln(dc + xc, sd, f + ".norm", 1e-12)
            if guided:
                xu = O._ln(du + xu, sd, f + ".norm", 1e-12)
        yc, yu = self.lin("last_layer.0.weight", "last_layer.0.bias", xc, xu, "x", head=True)
        yc = O._ln(F.gelu(yc), sd, "last_layer.2", 1e-12)
        yu = O._ln(F.gelu(yu), sd, "last_layer.2", 1e-12) if guided else None
        lc, lu = self.lin("prediction_layer.weight", "prediction_layer.bias", yc, yu, "x", head=True)
        shp = (B, cfg.seq + 1, cfg.splits, cfg.group_codes)
        lc = lc.reshape(shp)[:, :cfg.seq]
        lu = lu.reshape(shp)[:, :cfg.seq] if guided else None
        return lc, lu


EXACT = dict(x="exact", att="exact", h="exact", qkv="exact", p="exact", wt="exact", wh="exact", pair=True)
F16 = dict(x="f16", att="f16", h="f16", qkv="f16", p="f16", wt="f16", wh="f16", pair=True)


def cases(guided):
    c = [("single fp16, independent streams", {**F16, "pair": False}),
         ("differential form, fp16 weights (precision 1)", dict(F16)),
         ("+ exact trunk weights (the ideal precision 2)", {**F16, "wt": "exact"}),
         ("+ fp4 correction, 64-column block scales", {**F16, "wt": "corr4", "blk": 64}),
         ("+ fp4 correction, one scale per row", {**F16, "wt": "corr4", "blk": 0}),
         ("+ fp4 correction, 32-column block scales", {**F16, "wt": "corr4", "blk": 32}),
         ("+ fp4 correction, 64-column, weights per 128", {**F16, "wt": "corr4", "blk": 64, "wblk": 128}),
         ("+ fp4 correction, 64-column, scale skips top-1", {**F16, "wt": "corr4", "blk": 64, "skip": 1}),
         ("+ fp4 correction, 64-column, LayerNorm outputs skip top-1", {**F16, "wt": "corr4", "blk": 64, "skip": 1, "skip_x_only": True}),
         ("+ fp4 correction, 64-column, LayerNorm outputs skip top-1, w 128", {**F16, "wt": "corr4", "blk": 64, "skip": 1, "skip_x_only": True, "wblk": 128}),
         ("+ fp4 correction, 32-column, skip top-1, w 128", {**F16, "wt": "corr4", "blk": 32, "skip": 1, "wblk": 128}),
         ("+ fp4 correction, exact token operand", {**F16, "wt": "corr4", "blk": -1}),
         ("exact trunk weights + exact head weights", {**F16, "wt": "exact", "wh": "exact"}),
         ("exact weights, exact x", {**F16, "wt": "exact", "wh": "exact", "x": "exact"}),
         ("exact weights, exact att", {**F16, "wt": "exact", "wh": "exact", "att": "exact"}),
         ("exact weights, exact h", {**F16, "wt": "exact", "wh": "exact", "h": "exact"}),
         ("exact weights, exact qkv", {**F16, "wt": "exact", "wh": "exact", "qkv": "exact"}),
         ("exact weights, differential qkv", {**F16, "wt": "exact", "wh": "exact", "qkv": "diff"}),
         ("exact weights, exact p", {**F16, "wt": "exact", "wh": "exact", "p": "exact"}),
         ("exact weights, exact x att h", {**F16, "wt": "exact", "wh": "exact", "x": "exact", "att": "exact", "h": "exact"}),
         ("exact weights, exact qkv p", {**F16, "wt": "exact", "wh": "exact", "qkv": "exact", "p": "exact"}),
         ("coverage: qkv + up weights exact, exact head", {**F16, "wh": "exact", "wt_exact_for": ("in_proj", "net.0")}),
         ("coverage: out + down weights exact, exact head", {**F16, "wh": "exact", "wt_exact_for": ("out_proj", "net.2")}),
         ("coverage: up + down weights exact, exact head", {**F16, "wh": "exact", "wt_exact_for": ("net.0", "net.2")}),
         ("coverage: none (differential form), exact head", {**F16, "wh": "exact"}),
         ("only fp16 weights (trunk + head), rest exact", {**EXACT, "wt": "f16", "wh": "f16"}),
         ("only fp16 head weights, rest exact", {**EXACT, "wh": "f16"})]
    if not guided:
        c = [(n, o) for n, o in c if "differential" not in n and "independent" not in n] + [("single fp16", dict(F16))]
    return c


def cases_r6(guided):
    """Round 6: what closes the heavy-tailed runs.  ENG = the engine's default as built (precision 2): fp16 operands, differential form, MX-fp4 weight
    correction with 64-column token scales, head effectively exact (hi + lo inputs x hi + lo weights)."""
    ENG = {**F16, "wt": "corr4", "blk": 64, "wh": "exact"}
    A = lambda **kw: {"alo": kw}
    c = [("engine default (corr4, exact head)", dict(ENG)),
         ("exact trunk weights, exact head", {**ENG, "wt": "exact"}),
         ("corr4 two levels", {**ENG, "wlevels": 2}),
         ("corr6 weight error (e2m3)", {**ENG, "wt": "corr6"}),
         ("corr6 weight error, e2m3 tokens", {**ENG, "wt": "corr6", "tok6": True}),
         ("alo x e2m1", {**ENG, **A(x="e2m1")}),
         ("alo x top64", {**ENG, **A(x="top64")}),
         ("alo x f16", {**ENG, **A(x="f16")}),
         ("alo att e2m1", {**ENG, **A(att="e2m1")}),
         ("alo att f16", {**ENG, **A(att="f16")}),
         ("alo h e2m1", {**ENG, **A(h="e2m1")}),
         ("alo h f16", {**ENG, **A(h="f16")}),
         ("alo x att h e2m1", {**ENG, **A(x="e2m1", att="e2m1", h="e2m1")}),
         ("alo x att h e2m3", {**ENG, **A(x="e2m3", att="e2m3", h="e2m3")}),
         ("alo x att h f16", {**ENG, **A(x="f16", att="f16", h="f16")}),
         ("alo x att h e2m1 + corr4 two levels", {**ENG, "wlevels": 2, **A(x="e2m1", att="e2m1", h="e2m1")}),
         ("alo x att h e2m1 + corr6", {**ENG, "wt": "corr6", **A(x="e2m1", att="e2m1", h="e2m1")}),
         ("alo x att h f16 + exact weights", {**ENG, "wt": "exact", **A(x="f16", att="f16", h="f16")}),
         ("alo x att h f16 + exact weights + exact qkv p", {**ENG, "wt": "exact", "qkv": "exact", "p": "exact", **A(x="f16", att="f16", h="f16")})]
    if os.environ.get("EB_SET") == "2":
        X3 = A(x="e2m1", att="e2m1", h="e2m1")
        c = [("engine default (corr4, exact head)", dict(ENG)),
             ("tok6 only (e2m3 tokens, e2m1 weight error)", {**ENG, "tok6": True}),
             ("x top8 out of the token operand", {**ENG, "tok_top": 8}),
             ("x top32 out of the token operand", {**ENG, "tok_top": 32}),
             ("x att h top32 out of the token operand", {**ENG, "tok_top": 32, "tok_top_keys": ("x", "att", "h")}),
             ("32-column token scales", {**ENG, "blk": 32}),
             ("alo x att h e2m1", {**ENG, **X3}),
             ("alo x att h e2m1, x top8", {**ENG, **X3, "tok_top": 8}),
             ("alo x att h e2m1, x top32", {**ENG, **X3, "tok_top": 32}),
             ("alo x att h e2m1, x att h top32", {**ENG, **X3, "tok_top": 32, "tok_top_keys": ("x", "att", "h")}),
             ("alo x att h e2m1, 32-column token scales", {**ENG, **X3, "blk": 32}),
             ("alo x att h e2m1, x top32, weights per 128", {**ENG, **X3, "tok_top": 32, "wblk": 128}),
             ("alo x att h e2m1, x top32, weights per 32", {**ENG, **X3, "tok_top": 32, "wblk": 32})]
    if os.environ.get("EB_SET") == "3":
        X3 = A(x="e2m1", att="e2m1", h="e2m1")
        c = [("engine default (corr4, exact head)", dict(ENG)),
             ("weights per 32", {**ENG, "wblk": 32}),
             ("weights per 128", {**ENG, "wblk": 128}),
             ("alo x att h e2m1", {**ENG, **X3}),
             ("alo x att h e2m1, weights per 128", {**ENG, **X3, "wblk": 128}),
             ("alo x att h e2m1, weights per 32", {**ENG, **X3, "wblk": 32}),
             ("alo x att h e2m1, weights per 32, W values per 32", {**ENG, **X3, "wblk": 32, "w4blk": 32}),
             ("alo x att h e2m1, x top8, weights per 32", {**ENG, **X3, "tok_top": 8, "wblk": 32}),
             ("alo x att h e2m1, x top8, weights per 32, W values per 32", {**ENG, **X3, "tok_top": 8, "wblk": 32, "w4blk": 32}),
             ("alo x att h e2m1, x top32, weights per 32, W values per 32", {**ENG, **X3, "tok_top": 32, "wblk": 32, "w4blk": 32}),
             ("alo x e2m1 only, x top8, weights per 32, W values per 32", {**ENG, **A(x="e2m1"), "tok_top": 8, "wblk": 32, "w4blk": 32}),
             ("no alo, x top8, weights per 32", {**ENG, "tok_top": 8, "wblk": 32}),
             ]
    return c


def main():
    if os.environ.get("EB_STUDY") == "r6":
        global cases
        cases = cases_r6
    torch.set_num_threads(int(os.environ.get("THREADS", "8")))
    name = sys.argv[1] if len(sys.argv) > 1 else R.RUN_CFG5
    every = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    filt = sys.argv[3:]
    g = R.load_run(name)
    z = g["z"]
    cfg = synth.GenCfg(bits=g["bits"], splits=2)
    style = str(z["gen_style"]) if "gen_style" in z.files else "gaussian"
    sd = synth.make_generator_weights(cfg, seed=int(z["gen_seed"]), head_gain=float(z["head_gain"]), style=style)
    q, c = R.reference_noise(g, "cpu")
    scale, temp, mask_len = R.plan_of(g)
    S, B = g["steps"].shape[0], g["steps"].shape[1]
    guided_run = float(g["kw"]["guidance_scale"]) != 0.0
    steps = list(range(every // 2, S, every))
    if os.environ.get("EB_STEPS"):                       # e.g. EB_STEPS=1,3,5,7: these steps only
        steps = [int(t) for t in os.environ["EB_STEPS"].split(",")]
    C_ = g["C"]
    y = g["labels"]
    print(f"{name}: {S} steps, B = {B}, C = {C_}, steps used {steps}, scales {[round(scale[i], 2) for i in steps]}", flush=True)
    ref = {}
    exact = Emu(sd, cfg, EXACT)
    for i in steps:
        guided = guided_run and scale[i] != 0.0
        lc, lu = exact.forward(R.tokens_in(g, i), y, guided)
        ref[i] = lc + scale[i] * (lc - lu) if guided else lc
    for cname, o in cases(guided_run):
        if filt and not any(f in cname for f in filt):
            continue
        t0 = time.time()
        emu = Emu(sd, cfg, o)
        se = n = bad = tot = 0
        gse = gn = 0.0
        for i in steps:
            tin = R.tokens_in(g, i)
            guided = guided_run and scale[i] != 0.0
            lc, lu = emu.forward(tin, y, guided)
            L = lc + scale[i] * (lc - lu) if guided else lc
            msk = g["masks"][i]
            e = (L - ref[i])[msk]
            ec = e - e.mean(-1, keepdim=True)
            se += float(ec.pow(2).sum()); n += ec.numel()
            pred, _ = O.sample_step(lc, lu, scale[i], temp[i], q[i], c[i], tin, C_, torch.tensor(1.0), 512)
            bad += int((pred != g["steps"][i])[msk].sum()); tot += int(msk.sum())
            # what decides a flip: the error of the gap between the two best PERTURBED candidates (log-softmax - log q: argmax(p / q) of the draw)
            sref = torch.log_softmax(ref[i] / temp[i], -1) - torch.log(q[i].reshape(ref[i].shape))
            top2 = sref.topk(2, dim=-1).indices
            eg = (L - ref[i]).gather(-1, top2)
            ge = (eg[..., 0] - eg[..., 1])[msk]
            gse += float(ge.pow(2).sum()); gn += ge.numel()
        print(f"{cname:52s}: rms centred error of the sampled logits {math.sqrt(se / n):.5f}   rms error of the top-2 gap {math.sqrt(gse / gn):.5f}   "
              f"mismatch {bad}/{tot} = {bad / tot:.2e}   [{time.time() - t0:.0f} s]", flush=True)


if __name__ == "__main__":
    main()
