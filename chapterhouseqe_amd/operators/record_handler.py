"""RecordHandler: the exchange-side API the operator tasks call.

Reference: src/handlers/exchange_handlers/record_handler/record_handler.rs
  ExchangeRecord :48-52, initiate :79-125, next_record :127-214, complete_record :216-251,
  send_record_to_outbound_exchange :253-278, close.

Heartbeats: the reference spawns one RecordHeartbeatHandler task per pulled record (record_handler.rs:167-184) that
sends a RecordHeartbeatRequest every 100 ms until complete_record cancels it (heartbeat_handler.rs:78-82, 117-189); the
exchange requeues a reservation whose last heartbeat is older than 1 s (exchange_operator.rs:746-776).  Here one daemon
thread per RecordHandler renews every tracked record at the same 100 ms cadence, so a record that takes longer than a
second (first-call HIP initialisation, a large host batch over PCIe, a long group drain, a parquet write) is not handed
to a second operator instance while the first is still working on it.
"""
from __future__ import annotations

import dataclasses
import threading
import time
from typing import Any, List, Optional

from .exchange_operator import NONE_AVAILABLE, NONE_LEFT, ExchangeOperator


class RecordHandlerError(Exception):
    pass


@dataclasses.dataclass
class ExchangeRecord:
    record_id: int
    record: Any
    table_aliases: List[List[str]]


class RecordHandler:
    def __init__(self, operator_id: str, operator_instance_id: int, inbound_exchanges: List[ExchangeOperator],
                 outbound_exchange: Optional[ExchangeOperator], none_available_wait_time_s: float = 0.05):
        self.operator_id = operator_id
        self.operator_instance_id = operator_instance_id
        self.inbound_exchanges = inbound_exchanges
        self.outbound_exchange = outbound_exchange
        self.none_available_wait_time_s = none_available_wait_time_s      # 50 ms in the reference (:104)
        self.heartbeat_interval_s = 0.1                                   # heartbeat_handler.rs:80
        self.tracked_records = {}
        self._lock = threading.Lock()
        self._hb_stop = threading.Event()
        self._hb_thread: Optional[threading.Thread] = None
        self.heartbeat_errors: List[str] = []            # log of every failed beat (diagnostics only)
        self.max_request_runtime_errors = 25                # per record, heartbeat_handler.rs:83
        self._hb_error_counts = {}                          # record id -> failed beats of that record's heartbeat
        self._hb_failed: Optional[RecordHandlerError] = None

    # ---- periodic heartbeat of every tracked record (record_handler.rs:167-184, heartbeat_handler.rs:117-189) --------
    def _heartbeat_main(self) -> None:
        while not self._hb_stop.wait(self.heartbeat_interval_s):
            with self._lock:
                beats = list(self.tracked_records.items())
            for record_id, idx in beats:
                try:
                    self.inbound_exchanges[idx].heartbeat(self.operator_id, record_id)
                except Exception as e:  # noqa: BLE001 -- the reference logs and keeps beating, up to 25 errors PER RECORD
                    # (heartbeat_handler.rs:119-127: ReachedMaximumNumberOfRequestRuntimeErrorsAllowed ends the handler
                    # with an error); a record whose heartbeat has given up is no longer protected from being handed to
                    # a second instance, so the task must fail instead of carrying on: every later call raises
                    self.heartbeat_errors.append(repr(e))
                    with self._lock:
                        n = self._hb_error_counts.get(record_id, 0) + 1
                        self._hb_error_counts[record_id] = n
                        if n >= self.max_request_runtime_errors and self._hb_failed is None:
                            self._hb_failed = RecordHandlerError(
                                f"reached maximum number of request runtime errors allowed: {n} (heartbeat of record {record_id}; last: {e!r})")

    def _check_heartbeats(self) -> None:
        if self._hb_failed is not None:
            raise self._hb_failed

    def _track(self, record_id: int, exchange_idx: int) -> None:
        with self._lock:
            self.tracked_records[record_id] = exchange_idx
            if self._hb_thread is None or not self._hb_thread.is_alive():
                self._hb_stop.clear()
                self._hb_thread = threading.Thread(target=self._heartbeat_main, name=f"heartbeat-{self.operator_id}-{self.operator_instance_id}", daemon=True)
                self._hb_thread.start()
        # the first beat is immediate, like the reference's loop (request first, then sleep)
        self.inbound_exchanges[exchange_idx].heartbeat(self.operator_id, record_id)

    @staticmethod
    def initiate(op_in_config, inbound_exchanges: List[ExchangeOperator], outbound_exchange: Optional[ExchangeOperator]) -> "RecordHandler":
        return RecordHandler(op_in_config.operator_id, op_in_config.instance_id, inbound_exchanges, outbound_exchange)

    def next_record(self, max_wait_s: Optional[float] = None) -> Optional[ExchangeRecord]:
        """Pull the next record from the FIRST inbound exchange (a producer reads only its first inbound
        exchange, record_handler.rs:133-138). Returns None when the exchange has nothing left."""
        self._check_heartbeats()
        if not self.inbound_exchanges:
            raise RecordHandlerError("inbound exchanges is empty")
        ex = self.inbound_exchanges[0]
        deadline = None if max_wait_s is None else time.monotonic() + max_wait_s
        while True:
            got = ex.get_next_record(self.operator_id, self.operator_instance_id)
            if got == NONE_LEFT:
                return None
            if got == NONE_AVAILABLE:
                if deadline is not None and time.monotonic() > deadline:
                    return None
                time.sleep(self.none_available_wait_time_s)
                continue
            record_id, record, aliases = got
            self._track(record_id, 0)
            return ExchangeRecord(record_id, record, aliases)

    def try_next_record(self) -> Optional[ExchangeRecord]:
        """Non-blocking pull: a record that is queued right now, else None (nothing available yet, or nothing left).
        Not in the reference -- the GPU filter task uses it to drain the queue into one batch-group launch."""
        self._check_heartbeats()
        if not self.inbound_exchanges:
            raise RecordHandlerError("inbound exchanges is empty")
        ex = self.inbound_exchanges[0]
        got = ex.get_next_record(self.operator_id, self.operator_instance_id)
        if got == NONE_LEFT or got == NONE_AVAILABLE:
            return None
        record_id, record, aliases = got
        self._track(record_id, 0)
        return ExchangeRecord(record_id, record, aliases)

    def send_record_to_outbound_exchange(self, record_id: int, record: Any, table_aliases: List[List[str]]) -> None:
        self._check_heartbeats()
        if self.outbound_exchange is None:
            raise RecordHandlerError("outbound exchange is none")
        self.outbound_exchange.send_record(record_id, record, table_aliases)

    def complete_record(self, rec: ExchangeRecord) -> None:
        self._check_heartbeats()
        with self._lock:
            if rec.record_id not in self.tracked_records:
                raise RecordHandlerError(f"unable to find tracked record: {rec.record_id}")
            idx = self.tracked_records[rec.record_id]
        # ack first, then stop the heartbeat of this record (record_handler.rs:236-248)
        self.inbound_exchanges[idx].operator_completed_record_processing(self.operator_id, rec.record_id)
        with self._lock:
            self.tracked_records.pop(rec.record_id, None)
            self._hb_error_counts.pop(rec.record_id, None)

    def close(self) -> None:
        """cancel every heartbeat (the reference cancels the tracker's token and waits for the tasks)"""
        with self._lock:
            self.tracked_records.clear()
        self._hb_stop.set()
        t = self._hb_thread
        if t is not None and t.is_alive() and t is not threading.current_thread():
            t.join(timeout=2.0)
        self._hb_thread = None
