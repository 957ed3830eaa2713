"""CPU: engine/graph_decode.py::DecodeStager — the resident block tables with incremental append.  A random trace of decode
steps (sequences join, grow by a block now and then, sit steps out, leave, come back under a new block table; more live
sequences over time than table slots, so slots are recycled) is staged step by step; a numpy model of hx_stage_decode
applies every staging buffer to the "device" buffer, and after every step each row must read exactly what a full rebuild
(the reference's way: hydrainfer/engine/parameters_builder.py:46-97) would have given it."""
import random

import numpy as np
import pytest

from hydrainfer_amd._lib import HydraHipError
from hydrainfer_amd.engine.graph_decode import DecodeStager
from hydrainfer_amd.layer.causal_attention import decode_rank_descriptor


def apply_staging(dev: np.ndarray, st: np.ndarray, head_words: int) -> int:
    """What hx_stage_decode does (csrc/cache_ops.hip)."""
    dev[:head_words] = st[:head_words]
    n_runs, at, words = int(st[head_words]), head_words + 1, 0
    for _ in range(n_runs):
        off, cnt = int(st[at]), int(st[at + 1])
        assert off + cnt <= dev.size
        dev[off:off + cnt] = st[at + 2:at + 2 + cnt]
        at += 2 + cnt
        words += cnt
    return words


@pytest.mark.parametrize("seed", range(6))
def test_resident_tables_equal_a_full_rebuild(seed):
    rnd = random.Random(seed)
    max_batch, cap, bs, pad_block = 8, 12, 16, 999
    stg = DecodeStager(max_batch, cap, bs, pad_block, max_pos=4096, vocab=32000)
    dev = np.full(stg.total_words, -7, dtype=np.int32)
    st = np.zeros(stg.staging_words, dtype=np.int32)
    live = {}                 # sid -> (kv_len, table)
    next_sid, next_block = 1, 0
    appended = rebuilt = 0
    for step in range(300):
        # arrivals / departures / growth
        if len(live) < 20 and rnd.random() < 0.3:
            n0 = rnd.randint(1, 5)
            live[next_sid] = [rnd.randint((n0 - 1) * bs + 1, n0 * bs), list(range(next_block, next_block + n0))]
            next_block += n0
            next_sid += 1
        if live and rnd.random() < 0.1:
            live.pop(rnd.choice(list(live)))
        if live and rnd.random() < 0.05:       # the same sid with another table (cannot happen in the engine; must still be right)
            sid = rnd.choice(list(live))
            n0 = len(live[sid][1])
            live[sid][1] = list(range(next_block, next_block + n0))
            next_block += n0
        batch = rnd.sample(list(live), min(len(live), rnd.randint(1, max_batch))) if live else []
        if not batch:
            continue
        rows = []
        for r, sid in enumerate(batch):
            kv, tbl = live[sid]
            kv += 1
            if (kv + bs - 1) // bs > len(tbl) and len(tbl) < cap:
                tbl.append(next_block)
                next_block += 1
            kv = min(kv, len(tbl) * bs)
            live[sid][0] = kv
            pos = kv - 1
            tok = rnd.randint(1, 31999) if rnd.random() < 0.5 else -(rnd.randrange(max_batch) + 1)
            rows.append((tok, pos, tbl[pos // bs] * bs + pos % bs, kv, list(tbl), sid))
        B = (len(rows) + 3) // 4 * 4
        kv_max = stg.stage(st, rows, B)
        words = apply_staging(dev, st, stg.head_words)
        appended += words
        rebuilt += sum(len(r[4]) for r in rows)
        o = stg.off
        assert kv_max == max(r[3] for r in rows)
        kv_all = [r[3] for r in rows] + [1] * (B - len(rows))
        assert dev[o["kv_cu"]:o["kv_cu"] + B + 1].tolist() == [0] + list(np.cumsum(kv_all))
        assert dev[o["rank"]:o["rank"] + B + 1].tolist() == decode_rank_descriptor(kv_all)
        for r, row in enumerate(rows):
            assert dev[o["ids"] + r] == max(row[0], 0) and dev[o["src"] + r] == (-(row[0] + 1) if row[0] < 0 else -1)
            assert dev[o["pos"] + r] == row[1] and dev[o["slots"] + r] == row[2]
            start = stg.tables_off + int(dev[o["cu_blocks"] + r])
            assert dev[start:start + len(row[4])].tolist() == row[4], (step, r)
        for r in range(len(rows), B):          # padding rows: the scratch block
            start = stg.tables_off + int(dev[o["cu_blocks"] + r])
            assert dev[start] == pad_block and dev[o["slots"] + r] == pad_block * bs and dev[o["src"] + r] == -1
        # no two rows of a step share a table slot
        starts = [int(dev[o["cu_blocks"] + r]) for r in range(len(rows))]
        assert len(set(starts)) == len(starts) and 0 not in starts
    assert appended < 0.5 * rebuilt, (appended, rebuilt)       # the point of it: a step writes what is new, not every table


def test_stager_refuses_what_the_kernels_cannot_take():
    stg = DecodeStager(4, 4, 16, 9, max_pos=128, vocab=100)
    st = np.zeros(stg.staging_words, dtype=np.int32)
    with pytest.raises(HydraHipError):
        stg.stage(st, [(5, 128, 0, 129, [0] * 9, 1)], 4)        # position outside the rotary table
    with pytest.raises(HydraHipError):
        stg.stage(st, [(100, 3, 0, 4, [0], 1)], 4)              # token outside the vocabulary
    with pytest.raises(HydraHipError):
        stg.stage(st, [(5, 3, 0, 4, [0, 1, 2, 3, 4], 1)], 4)    # more blocks than the decoder was built for


@pytest.mark.gpu
def test_hx_stage_decode_applies_head_and_runs_like_the_numpy_model():
    """The device side of the stager: hx_stage_decode on pinned staging buffers == apply_staging, over a random trace."""
    import torch
    from hydrainfer_amd import _lib
    rnd = random.Random(3)
    max_batch, cap, bs = 16, 20, 16
    stg = DecodeStager(max_batch, cap, bs, 555, max_pos=4096, vocab=32000)
    host = np.full(stg.total_words, -7, dtype=np.int32)
    dev = torch.full((stg.total_words,), -7, dtype=torch.int32, device="cuda:0")
    staging = [torch.zeros(stg.staging_words, dtype=torch.int32).pin_memory() for _ in range(2)]
    tables = {sid: list(range(sid * 100, sid * 100 + rnd.randint(1, 6))) for sid in range(1, 40)}
    kv = {sid: len(t) * bs - rnd.randint(0, 15) for sid, t in tables.items()}
    for step in range(60):
        batch = rnd.sample(list(tables), rnd.randint(1, max_batch))
        rows = []
        for sid in batch:
            kv[sid] += 1
            if (kv[sid] + bs - 1) // bs > len(tables[sid]) and len(tables[sid]) < cap:
                tables[sid].append(sid * 100 + len(tables[sid]))
            kv[sid] = min(kv[sid], len(tables[sid]) * bs)
            pos = kv[sid] - 1
            rows.append((rnd.randint(1, 31999), pos, tables[sid][pos // bs] * bs + pos % bs, kv[sid], list(tables[sid]), sid))
        B = (len(rows) + 3) // 4 * 4
        st = staging[step % 2]
        torch.cuda.synchronize()
        stg.stage(st.numpy(), rows, B)
        apply_staging(host, st.numpy(), stg.head_words)
        _lib.check(_lib.lib().hx_stage_decode(dev.data_ptr(), dev.numel(), st.data_ptr(), stg.head_words, _lib.current_stream()), "stage")
        torch.cuda.synchronize()
        assert np.array_equal(dev.cpu().numpy(), host), step
