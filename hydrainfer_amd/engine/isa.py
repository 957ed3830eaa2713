"""Instruction set of the batch scheduler — mirror of hydrainfer/engine/isa.py:1-234.

A request is a doubly linked chain of instructions between two sentinel EmptyInstructions:
    ImageEmbed -> EPMigrate -> PullCache -> ImageEmbedFill -> PDMigrate -> PullCache
               -> TextFill (one per generated token) ... -> Empty
Fill carries the token ids, rotary positions and *virtual* cache ids of the tokens it feeds;
chunk_prefill(n) keeps the first n tokens in place and links the rest in as the next instruction."""
from typing import List, Optional, Tuple


class Instruction:
    next: Optional["Instruction"] = None
    prev: Optional["Instruction"] = None

    def insert_next(self, inst: "Instruction") -> None:
        inst.prev, inst.next = self, self.next
        self.next.prev = inst
        self.next = inst


class EmptyInstruction(Instruction):
    def __repr__(self):
        return "EM"


class Fill(Instruction):
    """Prefill (len(token_ids) > 1) or decode (== 1).  A decode Fill's token id is unknown when the
    request is created; the sampling Fill before it writes it through `sample_dst`."""

    def __init__(self, token_ids: Optional[List[int]], position_ids: List[int], cache_ids: List[int],
                 sample: bool, sample_dst: Optional["Fill"], hashes: Optional[List[int]]):
        self.token_ids = token_ids
        self.position_ids = position_ids
        self.cache_ids = cache_ids
        self.sample = sample
        self.sample_dst = sample_dst
        self.hashes = hashes
        self.is_chunked = False   # a chunk's sampled token is discarded (executor.py:163-165)

    def _split_common(self, rest: "Fill", chunk_size: int) -> None:
        self.insert_next(rest)
        self.token_ids = self.token_ids[:chunk_size]
        self.position_ids = self.position_ids[:chunk_size]
        self.cache_ids = self.cache_ids[:chunk_size]
        # the reference keeps sampling on the head chunk and throws the token away (isa.py:82-84)
        self.sample = True
        self.sample_dst = EmptyInstruction()
        self.is_chunked = True

    def chunk_prefill(self, chunk_size: int) -> None:
        raise NotImplementedError


class TextFill(Fill):
    def chunk_prefill(self, chunk_size: int) -> None:
        assert 0 < chunk_size < len(self.token_ids), f"invalid chunk prefill size {chunk_size}"
        # (the reference passes `hash=` here, isa.py:76, a TypeError; the intent — hand the hashes
        #  on, as ImageEmbedFill does — is what is implemented)
        rest = TextFill(self.token_ids[chunk_size:], self.position_ids[chunk_size:],
                        self.cache_ids[chunk_size:], self.sample, self.sample_dst, self.hashes)
        self._split_common(rest, chunk_size)

    def __repr__(self):
        return "TF"


class ImageEmbedFill(Fill):
    """Prefill whose image-token rows are overwritten with cached image embeddings:
    `image_token_mask[i]` marks token i as an image token, `image_token_cache_ids` are the
    virtual ids of those rows in the image cache (isa.py:96-141)."""

    def __init__(self, image_token_cache_ids: List[int], image_token_mask: List[bool],
                 token_ids: Optional[List[int]], position_ids: List[int], cache_ids: List[int],
                 sample: bool, sample_dst: Optional[Fill], hashes: Optional[List[int]]):
        super().__init__(token_ids, position_ids, cache_ids, sample, sample_dst, hashes)
        self.image_token_cache_ids = image_token_cache_ids
        self.image_token_mask = image_token_mask

    def chunk_prefill(self, chunk_size: int) -> None:
        assert 0 < chunk_size < len(self.token_ids), f"invalid chunk prefill size {chunk_size}"
        n_img = sum(self.image_token_mask[:chunk_size])
        rest = ImageEmbedFill(self.image_token_cache_ids[n_img:], self.image_token_mask[chunk_size:],
                              self.token_ids[chunk_size:], self.position_ids[chunk_size:],
                              self.cache_ids[chunk_size:], self.sample, self.sample_dst, self.hashes)
        self.image_token_cache_ids = self.image_token_cache_ids[:n_img]
        self.image_token_mask = self.image_token_mask[:chunk_size]
        self._split_common(rest, chunk_size)

    def __repr__(self):
        return "EF"


class ImageEmbed(Instruction):
    def __init__(self, pixel_values, cache_ids: List[int], images_size: List[Tuple[int, int]],
                 hashes: Optional[List[int]]):
        self.pixel_values = pixel_values
        self.cache_ids = cache_ids
        self.images_size = images_size
        self.hashes = hashes

    def __repr__(self):
        return "IE"


class MigrateRequest(Instruction):
    def __repr__(self):
        return "MR"


class EPMigrate(MigrateRequest):
    def __repr__(self):
        return "EPMR"


class PDMigrate(MigrateRequest):
    def __repr__(self):
        return "PDMR"


class PullCache(Instruction):
    src_node = None   # set by the receiving node (epdnode.py:407-410, `src_node_actor_handle`)
    hop = None        # 'ep' | 'pd': which migrate instruction this pull answers (set by the sender)

    def __repr__(self):
        return "PR"


class InstructionList:
    def __init__(self, head: Instruction, tail: Instruction, curr: Instruction):
        self.head, self.tail, self.curr = head, tail, curr

    def __iter__(self):
        node = self.head
        while node is not None:
            yield node
            node = node.next

    def __repr__(self):
        return "->".join(("*" if i is self.curr else "") + repr(i) for i in self)


class InstructionListBuilder:
    def __init__(self):
        self.head, self.tail = EmptyInstruction(), EmptyInstruction()
        self.head.next, self.tail.prev = self.tail, self.head

    def append(self, inst: Instruction) -> None:
        self.tail.prev.insert_next(inst)

    def build_instruction_list(self) -> InstructionList:
        return InstructionList(self.head, self.tail, self.head.next)
