"""LB_Init_Trainer (diffuser/libero/lb_train_utils.py:5-24): picks the trainer class and builds the CLIP tokenizer / text encoder.
The CLIP weights are fetched from the hub in the reference; without network access (or with V2A_TEXT_ENCODER=hash) a deterministic
token-hash encoder with the same call surface stands in, so the joint loop stays runnable (the text tower is out of scope:
SURVEY.md 8c item 5)."""
import os
import zlib
from types import SimpleNamespace
import torch
import torch.nn as nn


class HashTokenizer:
    """`tokenizer(list[str], return_tensors='pt', padding=True, truncation=True, max_length=128)` -> object with `.to()` that unpacks
    as **kwargs (input_ids, attention_mask), like a transformers BatchEncoding."""
    vocab = 49408

    class _Batch(dict):
        def to(self, device):
            return type(self)({k: v.to(device) for k, v in self.items()})

    def __call__(self, texts, return_tensors="pt", padding=True, truncation=True, max_length=128):
        rows = [[self.vocab - 2] + [zlib.crc32(w.encode()) % (self.vocab - 2) for w in t.split()][:max_length - 2] + [self.vocab - 1]
                for t in texts]
        L = max(len(r) for r in rows)
        ids = torch.tensor([r + [self.vocab - 1] * (L - len(r)) for r in rows], dtype=torch.int64)
        mask = torch.tensor([[1] * len(r) + [0] * (L - len(r)) for r in rows], dtype=torch.int64)
        return self._Batch(input_ids=ids, attention_mask=mask)


class HashTextEncoder(nn.Module):
    """input_ids [B,L] -> `.last_hidden_state` [B,L,512]: fixed (seeded, frozen) embedding table + position code."""

    def __init__(self, dim=512, vocab=HashTokenizer.vocab, max_len=128, seed=0):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        self.register_buffer("table", torch.randn(vocab, dim, generator=g) * 0.5)
        self.register_buffer("pos", torch.randn(max_len, dim, generator=g) * 0.1)
        self.eval()

    def forward(self, input_ids, attention_mask=None):
        h = self.table[input_ids] + self.pos[:input_ids.shape[1]][None]
        return SimpleNamespace(last_hidden_state=h)


def build_text_tower():
    if os.environ.get("V2A_TEXT_ENCODER", "") != "hash":
        try:
            from transformers import CLIPTextModel, CLIPTokenizer
            name = "openai/clip-vit-base-patch32"
            tok = CLIPTokenizer.from_pretrained(name, local_files_only=True)
            enc = CLIPTextModel.from_pretrained(name, local_files_only=True)
            enc.requires_grad_(False)
            enc.eval()
            return tok, enc
        except Exception as e:
            print(f"[ lb_train_utils ] CLIP weights unavailable ({type(e).__name__}); using the token-hash text encoder")
    return HashTokenizer(), HashTextEncoder()


class LB_Init_Trainer:
    def __init__(self, args) -> None:
        from .lb_online_trainer_v7 import LB_Online_Trainer_V7
        if getattr(args, 'trainer_type', None) != 'v7':
            raise NotImplementedError
        self.trainer_cls = LB_Online_Trainer_V7
        self.tokenizer, self.text_encoder = build_text_tower()
