"""Seeded regeneration of the reference's sample data sets.

The reference writes its samples with an UNSEEDED `rand::thread_rng` (src/bin/create_sample_data.rs:172-189), so
there is no canonical file to compare against; this module reproduces the schema and the distributions
(id: Int32 = 0..size-1, value1: Utf8 of `string_size` chars uniform in 'a'..='z', value2: Float32 ~ U[0,100);
all non-null; cut into `rows_per_file`-row files, create_sample_data.rs:113-155) from a fixed seed.
"""
from __future__ import annotations

from typing import List

import numpy as np
import pyarrow as pa

SCHEMA = pa.schema([pa.field("id", pa.int32(), False), pa.field("value1", pa.utf8(), False),
                    pa.field("value2", pa.float32(), False)])


def simple_table(size: int, string_size: int, seed: int = 0xC0FFEE) -> pa.RecordBatch:
    rng = np.random.default_rng(seed)
    ids = np.arange(size, dtype=np.int32)
    chars = rng.integers(ord("a"), ord("z") + 1, size=size * string_size, dtype=np.uint8)
    offsets = (np.arange(size + 1, dtype=np.int64) * string_size).astype(np.int32)
    value1 = pa.Array.from_buffers(pa.utf8(), size, [None, pa.py_buffer(offsets.tobytes()), pa.py_buffer(chars.tobytes())])
    value2 = (rng.random(size) * 100.0).astype(np.float32)
    return pa.RecordBatch.from_arrays([pa.array(ids), value1, pa.array(value2)], schema=SCHEMA)


def simple_batches(size: int = 100, string_size: int = 8, rows_per_file: int = 33, seed: int = 0xC0FFEE) -> List[pa.RecordBatch]:
    """simple: (100, 8, 33); simple_wide_string: (100, 100, 33); large_simple: (10_000, 8, 1000);
    huge_simple: (1_000_000, 8, 10_000) -- create_sample_data.rs:113-155."""
    table = simple_table(size, string_size, seed)
    return [table.slice(s, min(rows_per_file, size - s)) for s in range(0, size, rows_per_file)]
