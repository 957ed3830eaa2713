"""Request / chunk shapes of the chat-completions endpoint — the wire contract of
hydrainfer/entrypoint/api_protocol.py:6-52 (pydantic models there; plain dicts here) as
hydrainfer/entrypoint/api_server.py:89-152 uses them, and of the reference's own client
(benchmark/backend.py:13-64: it POSTs {model, messages: [{role, content: [text, image_url...]}], max_tokens,
temperature, stream: true} and reads `data: {json}` lines, `choices[0].delta.content`, until `data: [DONE]`).

Pinned to the reference by tests/golden/g13_api_protocol.json: the chunk strings the reference's pydantic models
produce (`model_dump_json(exclude_unset=True)`) and the prompt its LLaVA chat template renders, generated in the build
container by tests/golden/generate_goldens.py."""
import base64
import json
from dataclasses import dataclass
from typing import List, Optional

IMAGE_TOKEN = "<image>"            # hydrainfer/model/llava.py:195
CHUNK_OBJECT = "chat.completion.chunk"


class ProtocolError(ValueError):
    """A request the reference would reject (its asserts / pydantic validation -> HTTP 4xx / 500 there; 400 here)."""


@dataclass
class ChatRequest:
    model: str
    role: str
    text: str                       # message content with the image placeholders in front (api_server.py:62-79)
    image_png: Optional[bytes]      # decoded bytes of the (at most one) base64 PNG
    max_tokens: int
    stream: bool


def parse_chat_completion_request(body: dict) -> ChatRequest:
    """ChatCompletionRequest (api_protocol.py:22-27) + the checks of create_chat_completion (api_server.py:95-99) +
    _parse_content (:62-79): every image_url content becomes one `<image>` placeholder IN FRONT of the text, followed by
    a newline; only `data:image/png;base64,` URLs; one message, at most one image."""
    if not isinstance(body, dict):
        raise ProtocolError("the request body must be a JSON object")
    model, messages = body.get("model"), body.get("messages")
    if not isinstance(model, str):
        raise ProtocolError("model: string required")
    if not isinstance(messages, list) or len(messages) != 1:
        raise ProtocolError("only support one round chat")                       # api_server.py:95
    msg = messages[0]
    role = msg.get("role") if isinstance(msg, dict) else None
    if role not in ("user", "system", "assistant"):
        raise ProtocolError("messages[0].role must be user / system / assistant")
    content = msg.get("content")
    text, images = "", []
    if isinstance(content, str):
        text_content, image_content = content, ""
    elif isinstance(content, list):
        text_content = image_content = ""
        for c in content:
            kind = c.get("type") if isinstance(c, dict) else None
            if kind == "text":
                text_content += c.get("text") or ""
            elif kind == "image_url":
                url = (c.get("image_url") or {}).get("url", "")
                prefix, _, b64 = url.partition(",")
                if prefix != "data:image/png;base64":
                    raise ProtocolError(f"only support base 64 png image url but got {prefix}")   # api_server.py:76
                image_content += IMAGE_TOKEN
                images.append(b64)
            else:
                raise ProtocolError("content type must be 'text' or 'image_url'")
    else:
        raise ProtocolError("messages[0].content: string or list required")
    if len(images) > 1:
        raise ProtocolError(f"only support one image per request, but got {len(images)} images")   # api_server.py:98
    text = image_content + "\n" + text_content if isinstance(content, list) else text_content
    max_tokens = body.get("max_tokens", 16)
    if max_tokens is None:
        max_tokens = 16
    if not isinstance(max_tokens, int) or isinstance(max_tokens, bool) or max_tokens < 1:
        raise ProtocolError("max_tokens: positive integer required")
    try:
        png = base64.b64decode(images[0], validate=True) if images else None
    except Exception:
        raise ProtocolError("image_url: invalid base64")
    return ChatRequest(model=model, role=role, text=text, image_png=png, max_tokens=max_tokens,
                       stream=bool(body.get("stream", False)))


def render_llava_chat_prompt(role: str, content: str, bos_token: str = "<s>", eos_token: str = "</s>") -> str:
    """hydrainfer/model/chat_template/template_llava.jinja with add_generation_prompt, for the one-message
    conversations the endpoint accepts (api_server.py:95): bos + [system text] + 'USER: ' + content + newline +
    'ASSISTANT:' + newline (the template's own line break after the generation prompt)."""
    if role == "system":
        return bos_token + content + "ASSISTANT:\n"
    if role != "user":
        raise ProtocolError("Conversation roles must alternate user/assistant/user/assistant/...")
    return bos_token + "USER: " + content + "\n" + "ASSISTANT:\n"


def _dumps(obj) -> str:
    return json.dumps(obj, separators=(",", ":"), ensure_ascii=False)      # pydantic's model_dump_json: compact, UTF-8


def chat_stream_chunk(request_id: str, created: int, model: str, content: Optional[str], first: bool = False) -> str:
    """One `data:` line of the stream (api_server.py:119-146): the first chunk of a choice carries
    delta = {role: assistant, content: ""}, every other one delta = {content: text}; key order and `exclude_unset`
    as pydantic emits them."""
    delta = {"role": "assistant", "content": ""} if first else {"content": content}
    return "data: " + _dumps({"id": request_id, "object": CHUNK_OBJECT, "created": created, "model": model,
                              "choices": [{"index": 0, "delta": delta}]}) + "\n\n"


DONE = "data: [DONE]\n\n"
