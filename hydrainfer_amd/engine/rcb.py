"""Request control block and batch — mirror of hydrainfer/engine/rcb.py:8-71,
hydrainfer/request/request.py:6-39, hydrainfer/engine/metric.py:5-19 and
hydrainfer/engine/scenario.py:3-16."""
from dataclasses import dataclass, field
from enum import IntEnum
from typing import List, Optional, Tuple

from hydrainfer_amd.engine.isa import Instruction, InstructionList
from hydrainfer_amd.memory.token_cache import VirtualTokenCache


@dataclass
class SamplingParameters:
    max_tokens: int = 50
    eos_token_ids: List[int] = field(default_factory=list)


@dataclass
class RequestMetaData:
    n_images: int
    n_prompt_tokens: int
    n_text_tokens: int
    n_image_tokens: int


@dataclass
class RequestMetric:
    arrival_time: float = 0.0
    token_times: List[float] = field(default_factory=list)
    finished_time: float = 0.0
    # latency breakdown: [begin, end] stamps per phase
    encode_queueing: List[float] = field(default_factory=list)
    encode_execute: List[float] = field(default_factory=list)
    ep_transfer: List[float] = field(default_factory=list)
    prefill_queueing: List[float] = field(default_factory=list)
    prefill_execute: List[float] = field(default_factory=list)
    pd_transfer: List[float] = field(default_factory=list)
    decode_queueing: List[float] = field(default_factory=list)
    decode_execute: List[float] = field(default_factory=list)


class ScenarioType(IntEnum):
    Relaxed = 0
    Strict = 1


class ScenarioClassifier:
    def classify(self, n_text_tokens: int, n_output_tokens: int) -> ScenarioType:
        strict = n_text_tokens < 100 and n_output_tokens < 100
        return ScenarioType.Strict if strict else ScenarioType.Relaxed


# Bumped by whoever changes a live request behind the engine's back (a cancelled stream lowering max_tokens:
# entrypoint/api_server.py).  The executor's steady-state decode cohort (engine/executor.py) compares it instead of
# re-reading every request every step.
MUTATIONS = [0]


class OutputTokenProcessor:
    def append_token_id(self, token_id: int, is_last_token: bool = False) -> None:
        raise NotImplementedError

    def fail(self, exc: BaseException) -> None:
        """The request was terminated by the engine (a migration that failed twice, a node that died): the reference
        pushes a None token to the stream (hydrainfer/cluster/epdnode.py:440-442, `(request_id, None)`)."""
        self.append_token_id(None, True)


class LogOutputTokenProcessor(OutputTokenProcessor):
    def __init__(self):
        self.token_ids: List[int] = []

    def append_token_id(self, token_id: int, is_last_token: bool = False) -> None:
        self.token_ids.append(token_id)


class RequestControlBlock:
    def __init__(self):
        self.request_id = None
        self.sampling_params: Optional[SamplingParameters] = None
        self.request_metadata: Optional[RequestMetaData] = None
        self.instructions: Optional[InstructionList] = None
        self.virtual_kv_cache: Optional[VirtualTokenCache] = None
        self.virtual_image_cache: Optional[VirtualTokenCache] = None
        self.sid: int = -1
        self.output_token_processors: List[OutputTokenProcessor] = []
        self.output_token_ids: List[int] = []
        self.scenario_type: Optional[ScenarioType] = None
        self.metric = RequestMetric()
        self.eos_hit = False      # set when a token read back late (decode look-ahead) was end-of-sequence
        self.stream_rank: Optional[int] = None    # multi-process serving: the rank whose front end streams this request's
                                                  # tokens to a client (engine/distributed.py: the reference pushes them
                                                  # over zmq from every node, hydrainfer/engine/output_token_processor.py:92-140)
        self.path: List[int] = []                 # multi-process serving: the ranks that have owned this request, in order
        self.failed: Optional[str] = None         # why the engine terminated this request (EPDNode.terminate), if it did

    def current_instruction(self) -> Instruction:
        return self.instructions.curr

    def step(self) -> None:
        self.instructions.curr = self.instructions.curr.next

    def is_finished(self) -> bool:
        if self.instructions.curr is None or self.eos_hit:
            return True
        if len(self.output_token_ids) >= self.sampling_params.max_tokens:      # (>=: a cancelled stream lowers max_tokens, api_server.py)
            return True
        return bool(self.output_token_ids) and self.output_token_ids[-1] in self.sampling_params.eos_token_ids

    def release_instructions(self) -> None:
        """A finished request's chain is a doubly linked list — reference cycles that only the
        cyclic collector can free, and serving loops run with it off (serve.quiet_gc).  Cut the
        links so plain reference counting reclaims the instructions."""
        if self.instructions is None:
            return
        node = self.instructions.head
        while node is not None:
            nxt = node.next
            node.next = node.prev = None
            if hasattr(node, "sample_dst"):
                node.sample_dst = None
            node = nxt
        self.instructions.curr = None

    def register_output_token_processor(self, p: OutputTokenProcessor) -> None:
        self.output_token_processors.append(p)

    def __repr__(self):
        return f"rcb(sid={self.sid}, {self.instructions})"


class BatchRequest:
    def __init__(self, rcbs: Optional[List[RequestControlBlock]] = None):
        self.rcbs = rcbs if rcbs is not None else []

    def __len__(self):
        return len(self.rcbs)

    def __getitem__(self, idx: int) -> Tuple[RequestControlBlock, Instruction]:
        return self.rcbs[idx], self.rcbs[idx].instructions.curr

    def __iter__(self):
        # (a generator, not the index protocol: `for rcb, inst in batch` runs a dozen times per engine step)
        for rcb in self.rcbs:
            yield rcb, rcb.instructions.curr

    def append(self, rcb: RequestControlBlock) -> None:
        self.rcbs.append(rcb)

    def step(self) -> None:
        for rcb in self.rcbs:
            rcb.step()
