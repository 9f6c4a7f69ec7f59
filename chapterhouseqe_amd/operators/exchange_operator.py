"""In-process mirror of the reference's exchange: `RecordPool` + the request surface of `ExchangeOperator`.

Reference: src/handlers/operator_handler/operators/exchange_operator.rs
  RecordPool::{new :573-591, add_record :596-619, get_next_record :621-667, update_reserved_record_heartbeat
  :669-683, operator_completed_record_processing :684-739, requeue_reserved_records_with_stale_heartbeat :746-776}
  and ExchangeOperator's handling of SendRecord / GetNextRecord (NoneAvailable vs NoneLeft, :395-447).

The reference reaches the exchange through TCP/mpsc messages (out of scope); here the same state machine is
called directly, and thread-safely, by operator instances of one process (one instance per GPU).  Records are
handles: a pyarrow RecordBatch (host) or a DeviceRecordBatch (HBM) -- the pool never touches row data.
"""
from __future__ import annotations

import collections
import dataclasses
import threading
import time
from typing import Any, Deque, Dict, List, Optional, Tuple


class RecordPoolError(Exception):
    pass


@dataclasses.dataclass
class _RecordRef:
    id: int
    record: Any
    table_aliases: List[List[str]]
    processed_by_operators: List[str]


@dataclasses.dataclass
class _ReservedRecord:
    record_id: int
    operator_instance_id: int
    reserved_time: float
    last_heartbeat_time: Optional[float] = None


class _OperatorRecordQueue:
    def __init__(self, operator_id: str):
        self.operator_id = operator_id
        self.records_to_process: Deque[int] = collections.deque()
        self.records_reserved_by_operator: Dict[int, _ReservedRecord] = {}
        self.record_processing_metrics: Dict[int, int] = {}   # record id -> failure_count


class RecordPool:
    """Work-sharing queue: every record is enqueued once per *consumer operator*; each instance of that
    operator that asks gets a different record (exchange_operator.rs:621-667) -- which is what shards batches
    over one filter instance per GPU with no change to the pool."""

    def __init__(self, operator_ids: List[str], max_heartbeat_interval_s: float = 1.0):
        self.operator_ids = sorted(operator_ids)
        self.records: Dict[int, _RecordRef] = {}
        self.queues = [_OperatorRecordQueue(o) for o in self.operator_ids]
        self.max_heartbeat_interval_s = max_heartbeat_interval_s

    def _queue(self, operator_id: str) -> _OperatorRecordQueue:
        for q in self.queues:
            if q.operator_id == operator_id:
                return q
        raise RecordPoolError(f"operator does not exist: {operator_id}")

    def add_record(self, record_id: int, record: Any, table_aliases: List[List[str]]) -> bool:
        if record_id in self.records:       # records can only be added once
            return False
        self.records[record_id] = _RecordRef(record_id, record, table_aliases, [])
        for q in self.queues:
            q.records_to_process.append(record_id)
        return True

    def get_next_record(self, operator_id: str, operator_instance_id: int) -> Optional[Tuple[int, Any, List[List[str]]]]:
        q = self._queue(operator_id)
        if not q.records_to_process:
            return None
        record_id = q.records_to_process.popleft()
        rec = self.records[record_id]
        q.records_reserved_by_operator[record_id] = _ReservedRecord(record_id, operator_instance_id, time.monotonic())
        q.record_processing_metrics[record_id] = 0
        return record_id, rec.record, rec.table_aliases

    def update_reserved_record_heartbeat(self, operator_id: str, record_id: int) -> None:
        q = self._queue(operator_id)
        r = q.records_reserved_by_operator.get(record_id)
        if r is not None:
            r.last_heartbeat_time = time.monotonic()

    def operator_completed_record_processing(self, operator_id: str, record_id: int) -> None:
        q = self._queue(operator_id)
        if q.records_reserved_by_operator.pop(record_id, None) is None:
            raise RecordPoolError(f"reserved record instance missing for operator: {operator_id}")
        q.record_processing_metrics.pop(record_id, None)
        ref = self.records.get(record_id)
        if ref is None:
            raise RecordPoolError(f"record does not exist: {record_id}")
        if operator_id in ref.processed_by_operators:
            raise RecordPoolError(f"record {record_id} already processed by operator {operator_id}")
        ref.processed_by_operators.append(operator_id)
        if sorted(ref.processed_by_operators) == self.operator_ids:
            del self.records[record_id]     # every consumer operator is done with it

    def maintain(self) -> None:
        """requeue reservations whose heartbeat went stale, to the FRONT of the queue (:746-776)"""
        now = time.monotonic()
        for q in self.queues:
            stale = [rid for rid, r in q.records_reserved_by_operator.items()
                     if r.last_heartbeat_time is not None and now - r.last_heartbeat_time > self.max_heartbeat_interval_s]
            for rid in stale:
                del q.records_reserved_by_operator[rid]
                q.records_to_process.appendleft(rid)
                if rid not in q.record_processing_metrics:
                    raise RecordPoolError(f"record processing metrics do not exist: {rid} {q.operator_id}")
                q.record_processing_metrics[rid] += 1

    def outstanding(self, operator_id: str) -> int:
        q = self._queue(operator_id)
        return len(q.records_to_process) + len(q.records_reserved_by_operator)


NONE_AVAILABLE = "NoneAvailable"
NONE_LEFT = "NoneLeft"


class ExchangeOperator:
    """Single-instance exchange (requests/identify_exchange_requests.rs:264-273) between one producing operator
    and its consuming operators."""

    def __init__(self, exchange_id: str, outbound_producer_ids: List[str], max_heartbeat_interval_s: float = 1.0):
        self.id = exchange_id
        self._lock = threading.Lock()
        self._pool = RecordPool(outbound_producer_ids, max_heartbeat_interval_s)
        self.received_all_data_from_producers = False
        # the reference's RecordPoolMaintainer wakes every 100 ms (exchange_operator.rs:798-818); here the same sweep runs
        # inside get_next_record, at most once per interval (not on every pull: the sweep scans all reservations)
        self.maintain_interval_s = 0.1
        self._last_maintain = 0.0

    def send_record(self, record_id: int, record: Any, table_aliases: List[List[str]]) -> bool:
        with self._lock:
            return self._pool.add_record(record_id, record, table_aliases)

    def get_next_record(self, operator_id: str, operator_instance_id: int):
        with self._lock:
            now = time.monotonic()
            if now - self._last_maintain >= self.maintain_interval_s:
                self._last_maintain = now
                self._pool.maintain()
            got = self._pool.get_next_record(operator_id, operator_instance_id)
            if got is not None:
                return got
            if self.received_all_data_from_producers and self._pool.outstanding(operator_id) == 0:
                return NONE_LEFT
            return NONE_AVAILABLE

    def heartbeat(self, operator_id: str, record_id: int) -> None:
        with self._lock:
            self._pool.update_reserved_record_heartbeat(operator_id, record_id)

    def operator_completed_record_processing(self, operator_id: str, record_id: int) -> None:
        with self._lock:
            self._pool.operator_completed_record_processing(operator_id, record_id)

    def producers_completed(self) -> None:
        with self._lock:
            self.received_all_data_from_producers = True

    def num_records(self) -> int:
        with self._lock:
            return len(self._pool.records)
