"""Continuous-batching engine around the HIP hot path: instruction
chains, batch scheduler, per-step parameter builder, executors and the E/P/D node step loop.
Host logic mirrors hydrainfer/engine/* and hydrainfer/cluster/epdnode.py; everything that touches
a tensor goes through libhydra_hip."""
from hydrainfer_amd.engine.isa import (EmptyInstruction, EPMigrate, Fill, ImageEmbed, ImageEmbedFill,
                                       Instruction, InstructionList, InstructionListBuilder,
                                       MigrateRequest, PDMigrate, PullCache, TextFill)
from hydrainfer_amd.engine.rcb import (BatchRequest, LogOutputTokenProcessor, OutputTokenProcessor,
                                       RequestControlBlock, RequestMetaData, RequestMetric,
                                       SamplingParameters, ScenarioClassifier, ScenarioType)
from hydrainfer_amd.engine.scheduler import (BatchScheduler, BatchSchedulerConfig, BatchSchedulerContext,
                                             BatchSchedulerMetrics)
from hydrainfer_amd.engine.request_processor import InstructionCreator, TokenRequest
