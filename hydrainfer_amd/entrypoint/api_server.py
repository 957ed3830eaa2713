"""OpenAI-compatible streaming endpoint over the engine — hydrainfer/entrypoint/api_server.py:30-152.

The reference: FastAPI + uvicorn in front, requests handed to the cluster's Ray actors, generated text coming back over
a zmq PULL socket into per-request AsyncStreams.  Here: a stdlib asyncio HTTP/1.1 server (GET /health,
POST /v1/chat/completions with `stream: true`), the engine stepped by one background thread (EngineFrontend — the role
of AsyncEPDNode.loop, hydrainfer/cluster/epdnode.py:238-337), tokens handed from that thread to the request's
asyncio queue by an OutputTokenProcessor (the reference's OnlineStreamOutputTokenProcessor /
ZmqOutputTokenProcessor, hydrainfer/engine/output_token_processor.py:41-110).  Same request and chunk shapes
(api_protocol.py), so the reference's benchmark client (benchmark/backend.py:13-64) drives it unchanged.
Non-streaming requests are refused exactly where the reference raises (api_server.py:150)."""
import asyncio
import io
import itertools
import json
import queue
import threading
import time
import uuid
from typing import Callable, Optional

from hydrainfer_amd.engine.rcb import OutputTokenProcessor, SamplingParameters
from hydrainfer_amd.engine.request_processor import InstructionCreator, TokenRequest
from hydrainfer_amd.entrypoint import api_protocol as proto

ADMIT_PER_STEP = 8      # as engine/serve.py: arrivals taken between two engine steps


class StreamOutputTokenProcessor(OutputTokenProcessor):
    """Engine thread -> event loop: every sampled token's text goes into the request's asyncio queue; None ends it."""

    def __init__(self, loop: asyncio.AbstractEventLoop, tokenizer):
        self.loop, self.tokenizer = loop, tokenizer
        self.queue: asyncio.Queue = asyncio.Queue()
        self.n_tokens = 0
        self.on_end: Optional[Callable[["StreamOutputTokenProcessor"], None]] = None     # set by the front end: its registry of live streams
        self.ended = False

    def _put(self, item) -> None:
        try:
            self.loop.call_soon_threadsafe(self.queue.put_nowait, item)
        except RuntimeError:        # the loop is gone (server shut down while the engine drains)
            pass

    def _end(self) -> None:
        if not self.ended:
            self.ended = True
            if self.on_end is not None:
                self.on_end(self)

    def append_token_id(self, token_id: int, is_last_token: bool = False) -> None:
        if token_id is None:        # the engine terminated the request (EPDNode.terminate: the reference's None token)
            self.fail(RuntimeError("the request was terminated by the engine"))
            return
        self.n_tokens += 1
        self._put(self.tokenizer.decode(token_id))
        if is_last_token:
            self._put(None)
            self._end()

    def fail(self, exc: BaseException) -> None:
        if not self.ended:
            self._put(exc)
        self._end()


class EngineFrontend:
    """Owns the cluster (engine.node.LocalCluster or anything with add_request / step / idle) and steps it on ONE thread;
    requests enter through a thread-safe inbox between two steps."""

    def __init__(self, cluster, creator: InstructionCreator, device=None, on_step: Optional[Callable[[], None]] = None):
        self.cluster, self.creator, self.device, self.on_step = cluster, creator, device, on_step
        self.inbox: "queue.SimpleQueue" = queue.SimpleQueue()
        self.running = False
        self.thread: Optional[threading.Thread] = None
        self.error: Optional[BaseException] = None
        self.n_admitted = 0
        # every stream that has been submitted and has not ended: should the engine thread die, ALL of them are failed —
        # those still in the inbox and those already attached to a request (round-5 ADVICE: the latter used to wait for ever)
        self.live: dict = {}
        self.live_lock = threading.Lock()
        self.rcb_of: dict = {}          # processor id -> its request (local engine): what a cancelled stream ends

    def _forget(self, processor) -> None:
        with self.live_lock:
            self.live.pop(id(processor), None)
        self.rcb_of.pop(id(processor), None)

    def start(self) -> None:
        self.running = True
        self.thread = threading.Thread(target=self._loop, name="hx-engine", daemon=True)
        self.thread.start()

    def stop(self, timeout: float = 30.0) -> None:
        self.running = False
        if self.thread is not None:
            self.thread.join(timeout)

    def submit(self, request: TokenRequest, processor: StreamOutputTokenProcessor) -> None:
        processor.on_end = self._forget
        with self.live_lock:
            self.live[id(processor)] = processor
        self.inbox.put((request, processor))
        if self.error is not None:      # (checked AFTER the put: a loop that died in between drains nothing any more)
            processor.fail(RuntimeError(f"the engine thread has stopped: {self.error!r}"))

    def cancel(self, processor: StreamOutputTokenProcessor) -> None:
        """The client of this stream has gone: stop generating for it (its request ends at its next token and frees its
        blocks the ordinary way) instead of decoding to max_tokens for nobody."""
        self.inbox.put((None, processor))

    def _admit(self) -> int:
        n = 0
        while n < ADMIT_PER_STEP:
            try:
                request, processor = self.inbox.get_nowait()
            except queue.Empty:
                break
            n += 1
            if request is None:
                self._cancel(processor)
                continue
            if processor.ended:          # cancelled or failed before it got here
                continue
            try:
                self._start(request, processor)
                self.n_admitted += 1
            except Exception as e:
                processor.fail(e)
        return n

    def _start(self, request: TokenRequest, processor: StreamOutputTokenProcessor) -> None:
        rcb = self.creator.process(request)           # ValueError: prompt + max_tokens past the rotary table
        rcb.register_output_token_processor(processor)
        self.rcb_of[id(processor)] = rcb
        self.cluster.add_request(rcb)

    def _cancel(self, processor: StreamOutputTokenProcessor) -> None:
        rcb = self.rcb_of.get(id(processor))
        if rcb is not None and rcb.sampling_params is not None:
            rcb.sampling_params.max_tokens = max(1, len(rcb.output_token_ids))      # finished at its next look
            from hydrainfer_amd.engine import rcb as rcb_module
            rcb_module.MUTATIONS[0] += 1                                            # (the decode cohort must look again)
        processor.ended = True
        self._forget(processor)

    def _loop(self) -> None:
        try:
            if self.device is not None and getattr(self.device, "type", "") == "cuda":
                import torch
                torch.cuda.set_device(self.device)        # the current device is per thread
            while self.running:
                admitted = self._admit()
                worked = self.cluster.step()
                if self.on_step is not None:
                    self.on_step()
                if not worked and not admitted:
                    time.sleep(0.0005)                    # idle: poll the inbox ~2000 times a second
        except BaseException as e:          # a failing step must not leave streams waiting for ever
            self.error = e
            self.running = False
            with self.live_lock:
                stranded = list(self.live.values())
            for processor in stranded:       # in the inbox or attached to a request: every open stream ends now
                processor.fail(e)
            raise


class _RankCluster:
    """engine.distributed.RankEngine seen as the `cluster` an EngineFrontend steps."""

    def __init__(self, engine):
        self.engine = engine

    def step(self) -> int:
        self.engine.step()
        return 0 if (self.engine.node.idle() and not self.engine.outbox) else 1


class RankEngineFrontend(EngineFrontend):
    """The front end of a multi-process deployment (one engine node per GPU, engine/distributed.py; BASELINE configs[3] /
    [4]): it runs on ONE rank beside that rank's node; a request starts on the rank the reference's routing rule picks
    (cluster.py:178-184) and its tokens come back through the mailbox from whichever ranks sample them — the zmq PUSH /
    PULL pair of the reference (api_server.py:49-60, output_token_processor.py:92-140).  The other ranks run
    `serve_worker`."""

    def __init__(self, engine, creator: InstructionCreator, device=None):
        super().__init__(_RankCluster(engine), creator, device)
        self.engine = engine
        engine.creator = creator
        self._index = 0

    def _start(self, request: TokenRequest, processor: StreamOutputTokenProcessor) -> None:
        self.engine.submit(request, processor, self.creator, self._index)
        self._index += 1


def serve_worker(engine, creator: InstructionCreator, should_stop: Callable[[], bool]) -> None:
    """Step loop of a rank without a front end: takes `submit` / `migrate` / `pull` / `free` messages, posts tokens."""
    engine.creator = creator
    while not should_stop():
        engine.step()
        if engine.node.idle() and not engine.outbox:
            time.sleep(0.0005)


_REASONS = {200: "OK", 400: "Bad Request", 404: "Not Found", 405: "Method Not Allowed", 413: "Payload Too Large",
            500: "Internal Server Error", 501: "Not Implemented", 503: "Service Unavailable"}


class ApiServer:
    """APIServer of hydrainfer/entrypoint/api_server.py:30-156: /health and /v1/chat/completions."""

    def __init__(self, frontend: EngineFrontend, tokenizer, image_processor=None, host: str = "127.0.0.1", port: int = 8888,
                 max_body_bytes: int = 64 << 20, image_size=(336, 336)):
        self.frontend, self.tokenizer, self.image_processor = frontend, tokenizer, image_processor
        self.host, self.port, self.max_body_bytes, self.image_size = host, port, max_body_bytes, image_size
        self.server: Optional[asyncio.base_events.Server] = None
        self._ids = itertools.count(1)          # (request ids are drawn on executor threads)
        self.n_streams_open = 0

    # ------------------------------------------------------------------ request -> engine
    def _token_request(self, req: proto.ChatRequest) -> TokenRequest:
        prompt = self.tokenizer.apply_chat_template(req.role, req.text)          # api_server.py:99
        token_ids = self.tokenizer.encode(prompt)
        pixels, image_hash, size = None, 0, tuple(self.image_size)
        if req.image_png is not None:
            from PIL import Image
            try:
                image = Image.open(io.BytesIO(req.image_png))
                image.load()
            except Exception:
                raise proto.ProtocolError("image_url: not a decodable PNG")
            size = (image.height, image.width)
            if self.image_processor is None:
                raise proto.ProtocolError("this server was started without an image processor")
            pixels = self.image_processor.process(image)
            import xxhash
            image_hash = xxhash.xxh64(req.image_png).intdigest() >> 1             # content hash: the prefix cache's image key
        return TokenRequest(request_id=next(self._ids), token_ids=token_ids, pixel_values=pixels, image_size=size,
                            image_hash=image_hash, sampling_params=SamplingParameters(max_tokens=req.max_tokens))

    # ------------------------------------------------------------------ HTTP
    @staticmethod
    async def _send(writer: asyncio.StreamWriter, status: int, body: bytes = b"", content_type: str = "application/json") -> None:
        head = (f"HTTP/1.1 {status} {_REASONS.get(status, 'Error')}\r\ncontent-type: {content_type}\r\n"
                f"content-length: {len(body)}\r\nconnection: close\r\n\r\n")
        writer.write(head.encode() + body)
        await writer.drain()

    @staticmethod
    async def _chunk(writer: asyncio.StreamWriter, text: str) -> None:
        data = text.encode("utf-8")
        writer.write(f"{len(data):x}\r\n".encode() + data + b"\r\n")
        await writer.drain()

    async def _handle(self, reader: asyncio.StreamReader, writer: asyncio.StreamWriter) -> None:
        try:
            line = await reader.readline()
            parts = line.decode("latin-1").split()
            if len(parts) < 2:
                return
            method, path = parts[0], parts[1].split("?")[0]
            headers = {}
            while True:
                h = await reader.readline()
                if h in (b"\r\n", b"\n", b""):
                    break
                k, _, v = h.decode("latin-1").partition(":")
                headers[k.strip().lower()] = v.strip()
            if path == "/health":
                if method != "GET":
                    await self._send(writer, 405)
                elif self.frontend.error is not None:       # the engine thread is gone: nothing will ever be served
                    await self._send(writer, 503, json.dumps({"detail": f"engine stopped: {self.frontend.error!r}"[:300]}).encode())
                else:
                    await self._send(writer, 200)
                return
            if path != "/v1/chat/completions":
                await self._send(writer, 404, b'{"detail":"Not Found"}')
                return
            if method != "POST":
                await self._send(writer, 405, b'{"detail":"Method Not Allowed"}')
                return
            try:
                n = int(headers.get("content-length", "0") or 0)
                if n < 0:
                    raise ValueError
            except ValueError:
                await self._send(writer, 400, b'{"detail":"content-length must be a non-negative integer"}')
                return
            if n > self.max_body_bytes:
                await self._send(writer, 413, b'{"detail":"request body too large"}')
                return
            raw = await reader.readexactly(n)
            try:
                req = proto.parse_chat_completion_request(json.loads(raw))
                if not req.stream:
                    # api_server.py:149-150: `raise Exception('not support non stream chat completion')`
                    await self._send(writer, 501, json.dumps({"detail": "not support non stream chat completion"}).encode())
                    return
                # PNG decode + CLIP preprocessing + tokenizer: off the event loop, which is writing other streams' chunks
                token_request = await asyncio.get_running_loop().run_in_executor(None, self._token_request, req)
            except (proto.ProtocolError, json.JSONDecodeError, UnicodeDecodeError) as e:
                await self._send(writer, 400, json.dumps({"detail": str(e)}).encode())
                return
            await self._stream(writer, req, token_request)
        except (asyncio.IncompleteReadError, ConnectionError):
            pass
        finally:
            try:
                writer.close()
                await writer.wait_closed()
            except Exception:
                pass

    async def _stream(self, writer: asyncio.StreamWriter, req: proto.ChatRequest, token_request: TokenRequest) -> None:
        request_id = f"chatcmpl-{uuid.uuid4().hex[:22]}"          # the reference: shortuuid.random() (22 characters)
        created = int(time.time())
        processor = StreamOutputTokenProcessor(asyncio.get_running_loop(), self.tokenizer)
        self.frontend.submit(token_request, processor)
        writer.write(b"HTTP/1.1 200 OK\r\ncontent-type: text/event-stream; charset=utf-8\r\ncache-control: no-cache\r\n"
                     b"transfer-encoding: chunked\r\nconnection: close\r\n\r\n")
        await writer.drain()
        self.n_streams_open += 1
        first_sent = False
        try:
            while True:
                item = await processor.queue.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    # the reference's stream would end with the exception raised inside the generator: the client sees
                    # the connection close without [DONE]; here the reason travels in a last event first
                    await self._chunk(writer, "data: " + json.dumps({"error": {"message": str(item), "type": type(item).__name__}}) + "\n\n")
                    return
                if not first_sent:       # api_server.py:119-134: role chunk in front of the first text
                    await self._chunk(writer, proto.chat_stream_chunk(request_id, created, req.model, None, first=True))
                    first_sent = True
                if item:                 # api_server.py:135: empty pieces are not sent
                    await self._chunk(writer, proto.chat_stream_chunk(request_id, created, req.model, item))
            await self._chunk(writer, proto.DONE)
        except (ConnectionError, asyncio.CancelledError):
            self.frontend.cancel(processor)        # the client has gone: stop generating for it
            raise
        finally:
            self.n_streams_open -= 1
            try:
                writer.write(b"0\r\n\r\n")
                await writer.drain()
            except Exception:
                pass

    # ------------------------------------------------------------------ life cycle
    async def start(self) -> None:
        self.server = await asyncio.start_server(self._handle, self.host, self.port)
        self.port = self.server.sockets[0].getsockname()[1]        # port 0: the one the OS picked

    async def serve_forever(self) -> None:
        if self.server is None:
            await self.start()
        async with self.server:
            await self.server.serve_forever()

    async def close(self) -> None:
        if self.server is not None:
            self.server.close()
            await self.server.wait_closed()

    def run(self) -> None:
        """Blocking: engine thread + HTTP server (the reference's `APIServer.run`, api_server.py:154-156)."""
        self.frontend.start()
        try:
            asyncio.run(self.serve_forever())
        finally:
            self.frontend.stop()
