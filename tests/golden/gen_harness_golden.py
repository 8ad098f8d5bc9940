#!/usr/bin/env python3
"""Generate tests/golden/harness_golden.json by IMPORTING AND RUNNING the reference's harness code in this container:
`/root/reference/code/utils.py` (`computeTopNAccuracy`, :215-271) and `/root/reference/code/data.py` (`SeqRecDataset(mode="test")`,
:112-278; `BaseDataset.get_prefix_allowed_tokens_fn`, :84-104).  Only inputs and OUTPUTS are stored (the reference cannot travel).

Run (CPU, seconds):  PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_harness_golden.py

Harness notes (harness-side only, the reference files are untouched):
  * both modules `import ipdb` (a debugger that is not installed and never called on these paths): an empty module object is put
    in `sys.modules["ipdb"]` before the import;
  * no tokenizer ships offline: `get_prefix_allowed_tokens_fn(tokenizer)` is driven with a stub whose ids are the ones the
    reference's extended tokenizer assigns (32000 + rank of the token in the sorted new-token list, data.py:46-57 +
    finetune_llama.py:84; "Response:" -> the separator ids atspeed_amd.synth.RESPONSE_SEP stands for; eos = 2).
Fixtures:
  metrics      300 random (ground truth, prediction) users x 5 cut-offs -> the reference's precision / recall / NDCG / MRR
  synthetic    a small dataset written in the reference's file formats (index JSON + *_dict.npy), the inputs stored with it ->
               the reference's test split (history strings, labels, prompt text) and mask-function outputs
  real         Beauty / Games: counts, per-user digests of the reference's test split and the allowed-token dict (checked when
               the data files are present, i.e. in the build container)
"""
from __future__ import annotations

import hashlib
import json
import os
import sys
import tempfile
import types
from types import SimpleNamespace

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, "/root/reference/code")
sys.dont_write_bytecode = True
sys.modules.setdefault("ipdb", types.ModuleType("ipdb"))     # debugger import of utils.py:15 / data.py:17; never called here

import data as ref_data        # noqa: E402  the reference
import utils as ref_utils      # noqa: E402  the reference

RESPONSE_SEP = (13291, 29901)  # atspeed_amd.synth.RESPONSE_SEP
REF_DATA = "/root/reference/data"


class StubTokenizer:
    """ids of the reference's extended tokenizer for the strings the mask function asks about."""
    eos_token_id = 2

    def __init__(self, new_tokens):
        self.ids = {t: 32000 + r for r, t in enumerate(new_tokens)}

    def __call__(self, text):
        if text == "Response:":
            return {"input_ids": [1] + list(RESPONSE_SEP)}
        return {"input_ids": [1, self.ids[text]]}


def ds_args(data_path, dataset, **kw):
    base = dict(data_path=data_path, dataset=dataset, max_his_len=20, his_sep=", ", index_file=".LCRec-1e-3lr.json", add_prefix=False,
                llama=True, subseq=False)
    base.update(kw)
    return SimpleNamespace(**base)


def digest(strings) -> str:
    h = hashlib.sha256()
    for s in strings:
        h.update(s.encode("utf-8"))
        h.update(b"\x00")
    return h.hexdigest()


def split_record(ds, full: bool):
    """What the reference's test split holds: per user the label strings and the prompt text."""
    n = len(ds)
    rec = {"n_users": n, "labels_digest": digest("|".join(ds[i]["labels"]) for i in range(n)),
           "prompts_digest": digest(ds[i]["input_ids"] for i in range(n)),
           "history_len": [ds[i]["input_ids"].count("<a_") for i in range(n)] if full else None}
    pick = range(n) if full else sorted({0, 1, n // 2, n - 1})
    rec["users"] = {str(i): {"labels": list(ds[i]["labels"]), "text": ds[i]["input_ids"]} for i in pick}
    return rec


def mask_record(ds, sentences):
    tok = StubTokenizer(ds.get_new_tokens())
    ds.allowed_tokens = None
    fn = ds.get_prefix_allowed_tokens_fn(tok)
    outs = []
    for s in sentences:
        r = fn(0, torch.tensor(s, dtype=torch.long))
        outs.append(None if r is None else sorted(int(x) for x in r))
    return {"new_tokens_digest": digest(ds.get_new_tokens()), "n_new_tokens": len(ds.get_new_tokens()),
            "allowed_tokens": {str(i): sorted(int(x) for x in v) for i, v in ds.allowed_tokens.items()},
            "all_items": len(ds.get_all_items()), "fn": [[s, o] for s, o in zip(sentences, outs)]}


def metrics_fixture():
    rng = np.random.default_rng(2025)
    truth, pred = [], []
    for u in range(300):
        n_t = int(rng.integers(0, 4)) if u % 7 else 0                     # some users without ground truth (skipped by the reference)
        truth.append([int(x) for x in rng.choice(60, size=n_t, replace=False)])
        pred.append([int(x) for x in rng.permutation(60)[:20]])
    topN = [1, 3, 5, 10, 20]
    p, r, n, m = ref_utils.computeTopNAccuracy(truth, pred, topN)
    # string items, as the harness compares code tuples / strings rather than ints
    struth = [[f"i{x}" for x in t] for t in truth[:40]]
    spred = [[f"i{x}" for x in q] for q in pred[:40]]
    sp, sr, sn, sm = ref_utils.computeTopNAccuracy(struth, spred, [5, 20])
    return {"truth": truth, "pred": pred, "topN": topN, "precision": p, "recall": r, "ndcg": n, "mrr": m,
            "strings_first40": {"topN": [5, 20], "precision": sp, "recall": sr, "ndcg": sn, "mrr": sm}}


def synthetic_fixture():
    rng = np.random.default_rng(7)
    idx = {str(i): [f"<a_{rng.integers(12)}>", f"<b_{rng.integers(30)}>", f"<c_{rng.integers(30)}>", f"<d_{rng.integers(30)}>"] for i in range(120)}
    idx["120"] = idx["5"]                                                  # two items with one code tuple, as in the real files
    train = {u: [int(x) for x in rng.integers(0, 121, size=rng.integers(1, 30))] for u in range(25)}
    valid = {u: ([int(rng.integers(0, 121))] if u % 5 else []) for u in range(25)}
    test = {u: ([int(x) for x in rng.integers(0, 121, size=rng.integers(1, 3))] if u % 4 else []) for u in range(25)}
    out = {"index": idx, "train": train, "valid": valid, "test": test, "variants": {}}
    with tempfile.TemporaryDirectory() as d:
        os.makedirs(os.path.join(d, "toy"))
        json.dump(idx, open(os.path.join(d, "toy", "toy.LCRec-1e-3lr.json"), "w"))
        for name, dd in (("training", train), ("validation", valid), ("testing", test)):
            np.save(os.path.join(d, "toy", f"{name}_dict.npy"), np.array(dd, dtype=object), allow_pickle=True)
        for tag, kw in (("default", {}), ("his5_prefix", dict(max_his_len=5, add_prefix=True)), ("nolimit_sep", dict(max_his_len=-1, his_sep="; "))):
            ds = ref_data.SeqRecDataset(ds_args(d, "toy", **kw), mode="test")
            out["variants"][tag] = {"kw": kw, "split": split_record(ds, full=True)}
        ds = ref_data.SeqRecDataset(ds_args(d, "toy"), mode="test")
        toks = ds.get_new_tokens()
        ids = {t: 32000 + r for r, t in enumerate(toks)}
        item = [ids[t] for t in idx["7"]]
        prompt = [1, 450, 1404, 756] + list(RESPONSE_SEP)
        sentences = [prompt, prompt + item[:1], prompt + item[:2], prompt + item[:3], prompt + item, [1, 5, 6],
                     list(RESPONSE_SEP) + [9] + prompt + item[:2]]           # two separators: the LAST one counts
        out["mask"] = mask_record(ds, sentences)
    return out


def real_fixture(name):
    ds = ref_data.SeqRecDataset(ds_args(REF_DATA, name), mode="test")
    toks = ds.get_new_tokens()
    ids = {t: 32000 + r for r, t in enumerate(toks)}
    first = [ids[t] for t in ds.indices["0"]]
    prompt = [1, 450, 1404] + list(RESPONSE_SEP)
    rec = {"split": split_record(ds, full=False), "mask": mask_record(ds, [prompt, prompt + first[:2], prompt + first])}
    al = rec["mask"].pop("allowed_tokens")
    rec["mask"]["allowed_ranges"] = {i: [v[0], v[-1] + 1, len(v)] for i, v in al.items()}   # contiguous id ranges
    rec["mask"]["allowed_digest"] = digest(json.dumps(al[i]) for i in sorted(al))
    rec["mask"]["fn"] = [[s, (o if o is None or len(o) < 8 else [o[0], o[-1] + 1, len(o)])] for s, o in rec["mask"]["fn"]]
    return rec


def main():
    out = {"metrics": metrics_fixture(), "synthetic": synthetic_fixture(), "real": {n: real_fixture(n) for n in ("beauty", "games")}}
    path = os.path.join(HERE, "harness_golden.json")
    with open(path, "w") as f:
        json.dump(out, f, separators=(",", ":"))
    print("wrote", path, os.path.getsize(path), "bytes")
    print({n: (r["split"]["n_users"], r["mask"]["n_new_tokens"], r["mask"]["all_items"]) for n, r in out["real"].items()})


if __name__ == "__main__":
    main()
