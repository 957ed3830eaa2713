"""Wire / protocol half of SURVEY §8(f) rank 4: an OpenAI-compatible streaming endpoint in front of the engine —
hydrainfer/entrypoint/api_server.py:89-152 without FastAPI / zmq / Ray (stdlib asyncio)."""
from hydrainfer_amd.entrypoint.api_protocol import (ProtocolError, chat_stream_chunk, parse_chat_completion_request,
                                                    render_llava_chat_prompt)
from hydrainfer_amd.entrypoint.api_server import ApiServer, EngineFrontend, RankEngineFrontend, serve_worker
from hydrainfer_amd.entrypoint.tokenizer import HFTokenizer, SyntheticTokenizer

__all__ = ["ApiServer", "EngineFrontend", "RankEngineFrontend", "serve_worker", "HFTokenizer", "ProtocolError", "SyntheticTokenizer", "chat_stream_chunk",
           "parse_chat_completion_request", "render_llava_chat_prompt"]
