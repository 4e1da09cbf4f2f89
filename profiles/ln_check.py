import os, sys
import numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
from oracle import bert_oracle
from bergen_amd import BertEncoder, _lib
cfg = dict(vocab_size=3000, hidden_size=1024, num_hidden_layers=24, num_attention_heads=16, intermediate_size=4096,
           max_position_embeddings=256, type_vocab_size=2, layer_norm_eps=1e-12, hidden_act="gelu")
sd = bert_oracle.random_bert(cfg, seed=51)
bert_oracle.random_cls_head(cfg, seed=52, num_labels=1, sd=sd)
ids, mask, types = bert_oracle.random_batch(cfg, batch=5, max_len=90, seed=53)
pad = 128 - ids.shape[1]
ids, mask, types = (np.pad(a, ((0, 0), (0, pad))) for a in (ids, mask, types))
ref = bert_oracle.cross_encode(sd, cfg, ids, mask, types)
kw = {"input_ids": torch.from_numpy(ids), "attention_mask": torch.from_numpy(mask), "token_type_ids": torch.from_numpy(types)}
for small in (1, 0, 1, 0):
    _lib.init(0)
    _lib.set_option("ln_small", small)
    enc = BertEncoder(cfg, {k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, device=0)
    got = enc.classify(kw).cpu().numpy()
    hid = enc(**kw)[0].float().cpu().numpy()
    print("ln_small", small, "err", np.abs(got - ref).max(), "logits", got.ravel(), "ref", ref.ravel(), "finite", np.isfinite(hid).all(), flush=True)
    enc.close()
