"""The wire / protocol half of SURVEY §8(f) rank 4 (hydrainfer/entrypoint/api_server.py:89-152): chunk strings and the
rendered prompt pinned to the reference's own classes (tests/golden/g13_api_protocol.npz: its pydantic models'
`model_dump_json(exclude_unset=True)`, APIServer._parse_content, template_llava.jinja), and the server end to end on CPU
— a real engine (scheduler, block managers, instruction chains) over the oracle model, driven over HTTP by the
reference client's own request / parsing logic (benchmark/backend.py:13-64, restated here: the GPU box has no reference)."""
import asyncio
import base64
import io
import json
import os

import numpy as np
import pytest
import torch

from hydrainfer_amd.entrypoint import (ApiServer, EngineFrontend, ProtocolError, SyntheticTokenizer, chat_stream_chunk,
                                       parse_chat_completion_request, render_llava_chat_prompt)
from hydrainfer_amd.entrypoint.api_server import StreamOutputTokenProcessor
from tests.golden import cases as C

G13 = np.load(os.path.join(os.path.dirname(__file__), "golden", "g13_api_protocol.npz"))


def test_stream_chunks_equal_the_reference_models_byte_for_byte():
    a = C.API_CASE
    assert chat_stream_chunk(a["id"], a["created"], a["model"], None, first=True) == str(G13["first_chunk"])
    for piece, want in zip(a["pieces"], G13["content_chunks"].tolist()):
        assert chat_stream_chunk(a["id"], a["created"], a["model"], piece) == want
    # what the reference's client reads out of such a line (benchmark/backend.py:47-54)
    line = chat_stream_chunk(a["id"], a["created"], a["model"], " Hello").strip()
    assert line.startswith("data: ") and json.loads(line[len("data: "):])["choices"][0]["delta"].get("content") == " Hello"


def test_request_parsing_and_prompt_equal_the_reference():
    for name, msg in C.api_messages().items():
        req = parse_chat_completion_request({"model": "m", "messages": [msg], "max_tokens": 7, "temperature": 0.0, "stream": True})
        assert req.text == str(G13[f"parsed_{name}_content"])                      # _parse_content, api_server.py:62-79
        assert (req.image_png is not None) == bool(int(G13[f"parsed_{name}_n_images"]))
        assert render_llava_chat_prompt(req.role, req.text) == str(G13[f"prompt_{name}"])   # template_llava.jinja
        assert req.max_tokens == 7 and req.stream
    assert parse_chat_completion_request({"model": "m", "messages": [C.api_messages()["text_only"]]}).max_tokens == 16   # api_protocol.py:25
    url = C.api_messages()["image_text"]["content"][1]
    two = {"role": "user", "content": [{"type": "text", "text": "x"}, url, url]}
    for bad, what in (({"model": "m", "messages": [two]}, "one image"),
                      ({"model": "m", "messages": []}, "one round"),
                      ({"model": "m", "messages": [C.api_messages()["text_only"]] * 2}, "one round"),
                      ({"model": "m", "messages": [{"role": "user", "content": [{"type": "image_url", "image_url": {"url": "data:image/jpeg;base64,AAAA"}}]}]}, "png"),
                      ({"messages": [C.api_messages()["text_only"]]}, "model"),
                      ({"model": "m", "messages": [C.api_messages()["text_only"]], "max_tokens": 0}, "max_tokens")):
        with pytest.raises(ProtocolError, match=what):
            parse_chat_completion_request(bad)


def test_synthetic_tokenizer_is_deterministic_and_in_range():
    t = SyntheticTokenizer(image_token_id=32000)
    ids = t.encode(t.apply_chat_template("user", "<image>\nWhat is shown in this image?"))
    assert ids[0] == 1 and ids.count(32000) == 1 and ids == t.encode(t.apply_chat_template("user", "<image>\nWhat is shown in this image?"))
    assert all(i == 32000 or i == 1 or 1000 <= i < 31999 for i in ids) and len(ids) == 10
    assert t.decode(4711) == " <4711>"


# ---- end to end on CPU -------------------------------------------------------------------------------------------
def _png(seed):
    from PIL import Image
    rng = np.random.RandomState(seed)
    buf = io.BytesIO()
    Image.fromarray(rng.randint(0, 256, (56, 56, 3), dtype=np.uint8)).save(buf, format="PNG")
    return base64.b64encode(buf.getvalue()).decode()


def _payload(text, image_b64, max_tokens, stream=True):
    # benchmark/backend.py:17-41, field for field
    content = [{"type": "text", "text": text}]
    if image_b64 is not None:
        content.append({"type": "image_url", "image_url": {"url": f"data:image/png;base64,{image_b64}"}})
    return {"model": "tiny-llava", "messages": [{"role": "user", "content": content}], "max_tokens": max_tokens,
            "temperature": 0.0, "stream": stream}


async def _client_stream(base_url, payload):
    """benchmark/backend.py:43-56: POST, read lines, keep `data: ` ones, stop at [DONE], collect delta.content."""
    import httpx
    text, n_events, done = "", 0, False
    async with httpx.AsyncClient(timeout=None) as client:
        async with client.stream("POST", f"{base_url}/chat/completions", json=payload) as response:
            assert response.status_code == 200 and response.headers["content-type"].startswith("text/event-stream")
            async for line in response.aiter_lines():
                if not line or not line.startswith("data: "):
                    continue
                if line.strip() == "data: [DONE]":
                    done = True
                    break
                data = json.loads(line[len("data: "):])
                delta = data["choices"][0]["delta"].get("content")
                if isinstance(delta, str):
                    text += delta
                n_events += 1
    return text, n_events, done


def _tiny_server():
    from hydrainfer_amd.model.processor import ClipImageProcessor
    from tests.test_engine_e2e import creator, oracle_cluster
    cluster, _, _ = oracle_cluster(torch.float32, ["EPD"], chunked=True)
    tok = SyntheticTokenizer(image_token_id=C.TINY_IMAGE_TOKEN_ID, lo=3, hi=C.TINY_IMAGE_TOKEN_ID)
    front = EngineFrontend(cluster, creator())
    server = ApiServer(front, tok, ClipImageProcessor(size=56), host="127.0.0.1", port=0, image_size=(56, 56))
    return server, front, tok


def test_server_streams_what_the_engine_generates_to_the_reference_client():
    from tests.engine_util import run_trace
    from tests.test_engine_e2e import creator, oracle_cluster
    server, front, tok = _tiny_server()
    jobs = [("What is shown in this image?", _png(1), 5), ("Describe the weather. Briefly.", None, 3),
            ("What is shown in this image?", _png(2), 6), ("one two three four five six seven", _png(1), 4)]

    async def go():
        await server.start()
        front.start()
        base = f"http://127.0.0.1:{server.port}/v1"
        try:
            import httpx
            async with httpx.AsyncClient() as c:
                assert (await c.get(f"http://127.0.0.1:{server.port}/health")).status_code == 200
                assert (await c.get(f"http://127.0.0.1:{server.port}/nope")).status_code == 404
                r = await c.post(f"{base}/chat/completions", json=_payload("x", None, 3, stream=False))
                assert r.status_code == 501 and "non stream" in r.json()["detail"]            # api_server.py:150
                r = await c.post(f"{base}/chat/completions", json={"model": "m", "messages": []})
                assert r.status_code == 400 and "one round" in r.json()["detail"]
                r = await c.post(f"{base}/chat/completions", content=b"{not json", headers={"content-type": "application/json"})
                assert r.status_code == 400
            return await asyncio.gather(*[_client_stream(base, _payload(t, im, n)) for t, im, n in jobs])
        finally:
            front.stop()
            await server.close()
    got = asyncio.run(go())
    assert front.error is None and front.n_admitted == len(jobs) and server.n_streams_open == 0

    # the same requests straight through a fresh engine: the stream is the engine's tokens, piece by piece
    from hydrainfer_amd.entrypoint.api_protocol import parse_chat_completion_request as parse
    ref_server, _, _ = _tiny_server()
    reqs = [ref_server._token_request(parse(_payload(t, im, n))) for t, im, n in jobs]
    cluster, _, _ = oracle_cluster(torch.float32, ["EPD"], chunked=True)
    rcbs = run_trace(cluster, creator(), [(0, r) for r in reqs])
    for (text, n_events, done), rcb, (_, _, n) in zip(got, rcbs, jobs):
        assert done and len(rcb.output_token_ids) == n
        assert n_events == n + 1                                   # the role chunk + one chunk per token
        assert text == "".join(tok.decode(t) for t in rcb.output_token_ids)


def test_engine_failure_and_oversized_prompt_end_the_stream_loudly():
    server, front, tok = _tiny_server()

    async def go():
        await server.start()
        front.start()
        base = f"http://127.0.0.1:{server.port}/v1"
        try:
            import httpx
            # prompt + max_tokens past the rotary table: InstructionCreator.process raises (request_processor.py) -> an
            # error event, no [DONE]
            async with httpx.AsyncClient(timeout=None) as c:
                async with c.stream("POST", f"{base}/chat/completions", json=_payload("x", None, 5000)) as r:
                    lines = [l async for l in r.aiter_lines() if l.startswith("data: ")]
            assert len(lines) == 1 and "exceed max_position_embeddings" in json.loads(lines[0][6:])["error"]["message"]
            # the engine keeps serving afterwards
            return await _client_stream(base, _payload("still there?", None, 2))
        finally:
            front.stop()
            await server.close()
    text, n_events, done = asyncio.run(go())
    assert done and n_events == 3 and front.error is None


@pytest.mark.filterwarnings("ignore::pytest.PytestUnhandledThreadExceptionWarning")      # the injected failure is re-raised by the engine thread on purpose
def test_engine_thread_death_ends_every_open_stream_and_health_says_so():
    """Round-5 ADVICE: when the engine thread dies, the streams of requests already ADMITTED used to wait for ever (only
    the inbox was failed) and /health kept answering 200."""
    server, front, tok = _tiny_server()
    steps = [0]
    real_step = front.cluster.step

    def step():
        steps[0] += 1
        if steps[0] > 6 and front.n_admitted >= 2:
            raise RuntimeError("injected: the engine step blew up")
        return real_step()
    front.cluster.step = step

    async def one(base, payload):
        import httpx
        async with httpx.AsyncClient(timeout=20) as c:
            async with c.stream("POST", f"{base}/chat/completions", json=payload) as r:
                return [l async for l in r.aiter_lines() if l.startswith("data: ")]

    async def go():
        await server.start()
        front.start()
        base = f"http://127.0.0.1:{server.port}/v1"
        try:
            import httpx
            streams = await asyncio.gather(one(base, _payload("a b c d e f", _png(1), 200)), one(base, _payload("hello there", None, 200)))
            async with httpx.AsyncClient() as c:
                health = await c.get(f"http://127.0.0.1:{server.port}/health")
                late = await one(base, _payload("anyone?", None, 3))        # submitted to a dead engine: fails at once
            reader, writer = await asyncio.open_connection("127.0.0.1", server.port)     # (httpx will not send this itself)
            writer.write(b"POST /v1/chat/completions HTTP/1.1\r\nhost: x\r\ncontent-length: nonsense\r\n\r\n{}")
            await writer.drain()
            bad = (await reader.readline()).decode()
            writer.close()
            return streams, health, late, bad
        finally:
            front.stop()
            await server.close()
    streams, health, late, bad = asyncio.run(go())
    assert isinstance(front.error, RuntimeError) and not front.live
    for lines in streams + [late]:
        last = json.loads(lines[-1][6:])
        assert "error" in last and ("blew up" in last["error"]["message"] or "engine thread has stopped" in last["error"]["message"]), lines[-1]
        assert "data: [DONE]" not in lines
    assert health.status_code == 503 and "engine stopped" in health.json()["detail"]
    assert bad.startswith("HTTP/1.1 400"), bad


def test_a_client_that_disconnects_stops_its_request():
    """Round-5 ADVICE: a stream whose client went away kept decoding to max_tokens, holding its KV blocks."""
    server, front, tok = _tiny_server()

    async def go():
        await server.start()
        front.start()
        try:
            reader, writer = await asyncio.open_connection("127.0.0.1", server.port)
            body = json.dumps(_payload("tell me a very long story", None, 100)).encode()
            writer.write(b"POST /v1/chat/completions HTTP/1.1\r\nhost: x\r\ncontent-type: application/json\r\ncontent-length: "
                         + str(len(body)).encode() + b"\r\n\r\n" + body)
            await writer.drain()
            seen = b""
            while seen.count(b"data: ") < 3:          # a few tokens have come
                seen += await reader.read(4096)
            writer.close()                           # ... and the client hangs up
            node = front.cluster.nodes[0]
            for _ in range(400):                      # the engine drains: the request ends long before 100 tokens
                await asyncio.sleep(0.01)             # (idle() alone is racy from this thread: a step in flight holds its batch)
                if node.finished and front.cluster.idle() and server.n_streams_open == 0:
                    break
            return [len(r.output_token_ids) for r in node.finished], front.cluster.idle()
        finally:
            front.stop()
            await server.close()
    n_tokens, idle = asyncio.run(go())
    assert idle and len(n_tokens) == 1 and 3 <= n_tokens[0] < 60, n_tokens
    assert not front.live and front.error is None


def test_stream_processor_hands_tokens_across_threads():
    import threading

    async def go():
        p = StreamOutputTokenProcessor(asyncio.get_running_loop(), SyntheticTokenizer())
        th = threading.Thread(target=lambda: [p.append_token_id(7, False), p.append_token_id(9, True)])
        th.start()
        out = [await p.queue.get() for _ in range(3)]
        th.join()
        return out
    assert asyncio.run(go()) == [" <7>", " <9>", None]


# ---- one engine node per process (E / P / D), the front end on rank 0: tokens come back through the mailbox ----------
def _serve_rank(rank, roles, port, q):
    try:
        import time
        import torch.distributed as dist
        from hydrainfer_amd.engine import InstructionCreator
        from hydrainfer_amd.entrypoint import RankEngineFrontend, serve_worker
        from hydrainfer_amd.model.processor import ClipImageProcessor
        from tests.test_distributed_engine_cpu import BS, IMAGE_TOKEN, N_IMG, _build_engine, expected_tokens
        dist.init_process_group("gloo", rank=rank, world_size=len(roles), init_method=f"tcp://127.0.0.1:{port}")
        engine, kv, img = _build_engine(rank, roles, None)
        creator = InstructionCreator(IMAGE_TOKEN, N_IMG, BS)
        engine.open_mailbox("serve0")
        store = dist.distributed_c10d._get_default_store()
        dist.barrier()
        if rank != 0:
            serve_worker(engine, creator, lambda: store.check(["serve_stop"]))
            q.put((rank, "ok", len(engine.node.finished)))
            return
        tok = SyntheticTokenizer(image_token_id=IMAGE_TOKEN)
        front = RankEngineFrontend(engine, creator)
        server = ApiServer(front, tok, ClipImageProcessor(size=8), host="127.0.0.1", port=0, image_size=(8, 8))
        jobs = [("What is shown in this image?", _png(1), 5), ("Describe the weather. Briefly.", None, 3),
                ("What is shown in this image?", _png(2), 6), ("one two three four five six seven", _png(1), 4),
                ("text only again", None, 7), ("What is shown in this image?", _png(3), 2)]

        async def go():
            await server.start()
            front.start()
            try:
                return await asyncio.gather(*[_client_stream(f"http://127.0.0.1:{server.port}/v1", _payload(t, im, n)) for t, im, n in jobs])
            finally:
                front.stop()
                await server.close()
        got = asyncio.run(go())
        store.set("serve_stop", "1")
        assert front.error is None
        from hydrainfer_amd.entrypoint.api_protocol import parse_chat_completion_request as parse
        from hydrainfer_amd.engine import SamplingParameters, TokenRequest
        for (text, n_events, done), (t, im, n) in zip(got, jobs):
            ids = tok.encode(tok.apply_chat_template("user", parse(_payload(t, im, n)).text))
            want = expected_tokens(TokenRequest(0, ids, None, (8, 8), 0, SamplingParameters(max_tokens=n)))
            assert done and n_events == n + 1
            assert text == "".join(tok.decode(x) for x in want), (t, text)
        assert not engine.token_handlers
        q.put((rank, "ok", len(engine.node.finished)))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc(), 0))


@pytest.mark.parametrize("roles", [["EP", "D"], ["E", "P", "D"]], ids="-".join)
def test_endpoint_in_front_of_a_multi_process_engine(roles):
    """BASELINE configs[3] shape of deployment on CPU (gloo, stand-in models with closed-form tokens): the HTTP front end
    on rank 0, image requests entering at E, text-only ones forwarded to the P rank (cluster.py:178-184), the first token
    sampled on P and the rest on D — every one of them reaches the client's stream, in order."""
    import socket
    import torch.multiprocessing as mp
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_serve_rank, args=(r, roles, port, q)) for r in range(len(roles))]
    for p in procs:
        p.start()
    try:
        results = [q.get(timeout=240) for _ in procs]
    finally:
        for p in procs:
            p.join(timeout=30)
            if p.is_alive():
                p.kill()
    assert sorted(r[:2] for r in results) == [(r, "ok") for r in range(len(roles))], results
    finished = {r: n for r, _, n in results}
    assert finished[len(roles) - 1] == 6 and finished[0] == 0          # every request finished on the D rank
