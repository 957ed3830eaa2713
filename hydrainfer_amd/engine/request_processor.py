"""Request -> instruction chain — mirror of hydrainfer/engine/request_processor.py:47-173
(InstructionCreator) without the tokenizer / image processor (the caller passes token ids and
pre-processed pixel values; there are no checkpoints or tokenizers offline)."""
from dataclasses import dataclass
from typing import List, Tuple

from hydrainfer_amd.engine.isa import (EPMigrate, ImageEmbed, ImageEmbedFill, InstructionListBuilder,
                                       PDMigrate, PullCache, TextFill)
from hydrainfer_amd.engine.rcb import (RequestControlBlock, RequestMetaData, SamplingParameters,
                                       ScenarioClassifier)
from hydrainfer_amd.memory.shared_cache import compute_hash


@dataclass
class TokenRequest:
    """What is left of hydrainfer.request.Request once tokenizer and image processor have run."""
    request_id: int
    token_ids: List[int]                       # prompt, one image_token_id per image
    pixel_values: object = None                # (1, C, H, W) tensor or None
    image_size: Tuple[int, int] = (336, 336)   # (height, width) of the original image
    image_hash: int = 0
    sampling_params: SamplingParameters = None


class InstructionCreator:
    def __init__(self, image_token_id: int = 32000, n_image_tokens_per_image: int = 576,
                 block_size: int = 16, ignore_eos: bool = True, eos_token_id: int = 2,
                 max_position_embeddings: int = 4096):
        self.max_position_embeddings = max_position_embeddings
        self.image_token_id = image_token_id
        self.n_image_tokens_per_image = n_image_tokens_per_image
        self.block_size = block_size
        self.ignore_eos, self.eos_token_id = ignore_eos, eos_token_id
        self.scenario_classifier = ScenarioClassifier()

    def _insert_image_tokens(self, token_ids: List[int], image_hashes: List[int]):
        """Each image placeholder becomes n_image_tokens placeholders; the prefix hashes are taken
        over the prompt with the image's content hash standing in for the inserted placeholders
        (request_processor.py:64-81: the LAST placeholder keeps the token id itself)."""
        out, to_hash, image_id, total = [], [], -1, 0
        for t in token_ids:
            if t == self.image_token_id:
                image_id += 1
                n = self.n_image_tokens_per_image
                total += n
                out.extend([self.image_token_id] * (n - 1))
                to_hash.extend([image_hashes[image_id]] * (n - 1))
            out.append(t)
            to_hash.append(t)
        return compute_hash(token_ids=to_hash, block_size=self.block_size, prefix=-1), out, total

    def process(self, request: TokenRequest) -> RequestControlBlock:
        rcb = RequestControlBlock()
        rcb.request_id = request.request_id
        sp = request.sampling_params or SamplingParameters()
        rcb.sampling_params = SamplingParameters(sp.max_tokens, list(sp.eos_token_ids))
        if not self.ignore_eos:
            rcb.sampling_params.eos_token_ids.append(self.eos_token_id)

        has_image = request.pixel_values is not None
        image_hashes = [request.image_hash] if has_image else []
        n_images = request.token_ids.count(self.image_token_id)
        hashes, token_ids, n_image_tokens = self._insert_image_tokens(request.token_ids, image_hashes)
        n_prompt = len(token_ids)
        # position ids run to n_prompt + max_tokens - 2; the rotary table (cos_sin) has
        # max_position_embeddings rows and the kernels index it unchecked
        if n_prompt + rcb.sampling_params.max_tokens - 1 > self.max_position_embeddings:
            raise ValueError(f"request {request.request_id}: {n_prompt} prompt tokens + "
                             f"{rcb.sampling_params.max_tokens} generated exceed max_position_embeddings "
                             f"= {self.max_position_embeddings}")
        token_ids = token_ids + [-1] * (rcb.sampling_params.max_tokens - 1)   # filled in while decoding
        mask = [t == self.image_token_id for t in token_ids]
        ids = list(range(len(token_ids)))        # position ids == virtual cache ids

        b = InstructionListBuilder()
        if has_image:
            image_cache_ids = list(range(n_image_tokens))
            b.append(ImageEmbed(request.pixel_values, image_cache_ids, [request.image_size], image_hashes))
            b.append(EPMigrate())
            b.append(PullCache())
            prefill = ImageEmbedFill(image_cache_ids, mask[:n_prompt], token_ids[:n_prompt], ids[:n_prompt],
                                     ids[:n_prompt], True, None, hashes)
        else:
            prefill = TextFill(token_ids[:n_prompt], ids[:n_prompt], ids[:n_prompt], True, None, hashes)
        b.append(prefill)
        b.append(PDMigrate())
        b.append(PullCache())
        last = prefill
        for i in range(n_prompt, len(token_ids)):
            decode = TextFill(token_ids[i:i + 1], ids[i:i + 1], ids[i:i + 1], True, None, None)
            b.append(decode)
            last.sample_dst = decode
            last = decode

        rcb.instructions = b.build_instruction_list()
        rcb.request_metadata = RequestMetaData(n_images=n_images, n_prompt_tokens=n_prompt,
                                               n_image_tokens=n_image_tokens,
                                               n_text_tokens=n_prompt - n_image_tokens)
        rcb.scenario_type = self.scenario_classifier.classify(
            n_text_tokens=rcb.request_metadata.n_text_tokens, n_output_tokens=rcb.sampling_params.max_tokens)
        return rcb
