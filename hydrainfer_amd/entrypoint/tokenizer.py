"""Tokenizer seam of the endpoint — hydrainfer/model/model_factory.py:56-62 (`Tokenizer.encode / decode /
apply_chat_template`) and hydrainfer/model/llava.py:143-176.

There are no tokenizer files offline, so two implementations: HFTokenizer wraps transformers' AutoTokenizer exactly as
the reference's LlavaTokenizer does (used when a checkpoint directory exists), SyntheticTokenizer is a deterministic
stand-in for benchmarks on random weights: words hash to ids, ids print as `<id>` pieces."""
import re
from typing import List

from hydrainfer_amd.entrypoint.api_protocol import IMAGE_TOKEN, render_llava_chat_prompt


class SyntheticTokenizer:
    """text -> ids: `<image>` -> image_token_id, `<s>` -> bos id, every other whitespace-separated piece -> a stable
    64-bit FNV-1a hash folded into [lo, hi).  ids -> text: ' <id>'.  Same interface as the reference's Tokenizer."""

    def __init__(self, image_token_id: int = 32000, lo: int = 1000, hi: int = 31999, bos_token_id: int = 1, eos_token_id: int = 2):
        self.image_token_id, self.lo, self.hi = image_token_id, lo, hi
        self.bos_token, self.eos_token = "<s>", "</s>"
        self.bos_token_id, self.eos_token_id = bos_token_id, eos_token_id

    @staticmethod
    def _fnv(piece: str) -> int:
        h = 0xcbf29ce484222325
        for b in piece.encode("utf-8"):
            h = ((h ^ b) * 0x100000001b3) & 0xFFFFFFFFFFFFFFFF
        return h

    def encode(self, prompt: str) -> List[int]:
        ids = []
        for piece in re.findall(r"<image>|<s>|</s>|\S+", prompt):
            if piece == IMAGE_TOKEN:
                ids.append(self.image_token_id)
            elif piece == self.bos_token:
                ids.append(self.bos_token_id)
            elif piece == self.eos_token:
                ids.append(self.eos_token_id)
            else:
                ids.append(self.lo + self._fnv(piece) % (self.hi - self.lo))
        return ids

    def decode(self, token_id: int) -> str:
        return f" <{int(token_id)}>"

    def apply_chat_template(self, role: str, content: str) -> str:
        return render_llava_chat_prompt(role, content, self.bos_token, self.eos_token)


class HFTokenizer:
    """hydrainfer/model/llava.py:143-176 over a local checkpoint directory (transformers.AutoTokenizer)."""

    def __init__(self, path: str):
        from transformers import AutoTokenizer
        self.tokenizer = AutoTokenizer.from_pretrained(path)
        self.bos_token, self.eos_token = self.tokenizer.bos_token, self.tokenizer.eos_token
        self.eos_token_id = self.tokenizer.eos_token_id

    def encode(self, prompt: str) -> List[int]:
        return self.tokenizer.encode(prompt, add_special_tokens=False)

    def decode(self, token_id: int) -> str:
        # (U+2581, not an ASCII underscore: a piece that starts a word gets its blank back — llava.py:160-166)
        if self.tokenizer.convert_ids_to_tokens([token_id])[0].startswith("▁"):
            return " " + self.tokenizer.decode([token_id])
        return self.tokenizer.decode([token_id])

    def apply_chat_template(self, role: str, content: str) -> str:
        return render_llava_chat_prompt(role, content, self.bos_token, self.eos_token)
