"""Operator tasks and the plugin API they register through.

Reference (OPS = src/handlers/operator_handler/operators):
  TaskBuilder trait                         OPS/traits.rs:22-36
  OperatorTaskRegistry                      OPS/operator_task_registry.rs:40-162
  FilterConfig / FilterTask / FilterTaskBuilder        OPS/filter_tasks/{config.rs, filter_task.rs:30-199}
  MaterializeFilesConfig / MaterializeFilesTask / ...  OPS/materialize_tasks/{config.rs, materialize_files_task.rs:30-230}
  OperatorTask::{Filter{expr}, MaterializeFiles{data_format, fields}}  src/planner/physical_planner.rs:58-65

The hot loops call `record_utils.filter_record` / `project_record` exactly where the reference does
(filter_task.rs:99, materialize_files_task.rs:110); here those are the HIP kernels.  The control plane around
them (message router, pipes, TCP) is out of scope: `build` receives the exchanges directly.
"""
from __future__ import annotations

import abc
import dataclasses
import os
import uuid
from typing import Any, Callable, List, Optional, Sequence

from .. import record_utils
from .. import sqlast as A
from .exchange_operator import ExchangeOperator
from .record_handler import RecordHandler


# ---- planner types the tasks consume ------------------------------------------------------------------
@dataclasses.dataclass(frozen=True)
class FilterOperatorTask:
    """planner::OperatorTask::Filter { expr }"""
    expr: A.Expr

    def task_name(self) -> str:
        return "filter"


@dataclasses.dataclass(frozen=True)
class MaterializeFilesOperatorTask:
    """planner::OperatorTask::MaterializeFiles { data_format, fields }"""
    data_format: str
    fields: Sequence[A.SelectItem]

    def task_name(self) -> str:
        return "materialize_files"


@dataclasses.dataclass(frozen=True)
class ReadFilesOperatorTask:
    """planner::OperatorTask::TableFunc { alias, func_name: "read_files", args, max_rows_per_batch } with the path argument
    already parsed (ReadFilesConfig::parse_config, read_files_task.rs:66-107)"""
    path: str                       # glob below the connection's root
    alias: Optional[str] = None
    max_rows_per_batch: int = 10_000   # physical_planner.rs:323

    def task_name(self) -> str:
        return "read_files"


@dataclasses.dataclass
class OperatorInstanceConfig:
    """operator_handler_state.rs:28-35 (fields the tasks use)"""
    instance_id: int
    operator_id: str
    query_id: int
    task: Any
    device_id: int = 0


# ---- plugin API ------------------------------------------------------------------------------------------
class TaskBuilder(abc.ABC):
    """OPS/traits.rs:22-36.  `build` returns a callable that runs the task to completion and returns None on
    success or the error (the reference sends that through a oneshot channel)."""

    @abc.abstractmethod
    def build(self, op_in_config: OperatorInstanceConfig, inbound_exchanges: List[ExchangeOperator],
              outbound_exchange: Optional[ExchangeOperator]) -> Callable[[], Optional[Exception]]:
        ...


class OperatorTaskRegistryError(Exception):
    pass


class OperatorTaskRegistry:
    def __init__(self):
        self.filter_task: Optional[TaskBuilder] = None
        self.materialize_files_task: Optional[TaskBuilder] = None
        self.materialize_data_formats: List[str] = []
        self.table_func_tasks: dict = {}

    def add_filter_task_builder(self, builder: TaskBuilder) -> "OperatorTaskRegistry":
        if self.filter_task is not None:
            raise OperatorTaskRegistryError("filter task builder already set")
        self.filter_task = builder
        return self

    def add_materialize_files_builder(self, builder: TaskBuilder, data_formats: List[str]) -> "OperatorTaskRegistry":
        if self.materialize_files_task is not None:
            raise OperatorTaskRegistryError("materialize file task builder already set")
        self.materialize_files_task = builder
        self.materialize_data_formats = list(data_formats)
        return self

    def add_table_func_task_builder(self, func_name: str, builder: TaskBuilder) -> "OperatorTaskRegistry":
        """operator_task_registry.rs:35-49 (the syntax validator is the planner's business: out of scope)"""
        if func_name in self.table_func_tasks:
            raise OperatorTaskRegistryError(f"table func task builder for {func_name} already set")
        self.table_func_tasks[func_name] = builder
        return self

    def find_task_builder(self, task) -> Optional[TaskBuilder]:
        if isinstance(task, ReadFilesOperatorTask):
            return self.table_func_tasks.get("read_files")
        if isinstance(task, FilterOperatorTask):
            return self.filter_task
        if isinstance(task, MaterializeFilesOperatorTask):
            if task.data_format in self.materialize_data_formats:
                return self.materialize_files_task
        return None


# ---- filter --------------------------------------------------------------------------------------------------
@dataclasses.dataclass(frozen=True)
class FilterConfig:
    expr: A.Expr

    @staticmethod
    def try_from(op_in_config: OperatorInstanceConfig) -> "FilterConfig":
        if not isinstance(op_in_config.task, FilterOperatorTask):
            raise ValueError("operator instance config is not a filter task")
        return FilterConfig(op_in_config.task.expr)


class FilterTask:
    """filter_task.rs:65-142.  `group_size` > 1 is the GPU extension: the task drains up to that many queued records
    of one schema and filters them with ONE kernel launch (`record_utils.filter_records`); every output still
    carries its input record id and is acked individually, so the exchange protocol is unchanged."""

    def __init__(self, op_in_config: OperatorInstanceConfig, filter_config: FilterConfig,
                 inbound_exchanges, outbound_exchange, filter_fn=None, ctx=None, group_size: int = 1):
        self.operator_instance_config = op_in_config
        self.filter_config = filter_config
        self.inbound_exchanges = inbound_exchanges
        self.outbound_exchange = outbound_exchange
        self._filter = filter_fn
        self._ctx = ctx
        self.group_size = max(1, int(group_size))
        self.records_processed = 0
        self.rows_in = 0
        self.rows_out = 0
        self.group_calls = 0

    def _context(self):
        if self._ctx is None:
            self._ctx = record_utils.Context(self.operator_instance_config.device_id)
        return self._ctx

    def _filter_record(self, rec, aliases):
        if self._filter is not None:
            return self._filter(rec, aliases, self.filter_config.expr)
        return record_utils.filter_record(rec, aliases, self.filter_config.expr, ctx=self._context())

    def _filter_group(self, group):
        """one library call for records that share schema, residency and table aliases; else record by record"""
        first = group[0]
        same = self._filter is None and len(group) > 1 and all(
            g.table_aliases == first.table_aliases and type(g.record) is type(first.record) and
            _schema_of(g.record) == _schema_of(first.record) for g in group[1:])
        if not same:
            return [self._filter_record(g.record, g.table_aliases) for g in group]
        self.group_calls += 1
        return record_utils.filter_records([g.record for g in group], first.table_aliases, self.filter_config.expr,
                                           ctx=self._context())

    def async_main(self) -> None:
        """filter_task.rs:65-142: pull -> filter -> push (same record id) -> ack"""
        rec_handler = RecordHandler.initiate(self.operator_instance_config, self.inbound_exchanges, self.outbound_exchange)
        try:   # an error ends the task (filter_task.rs `?`): its heartbeats must stop with it, so the exchange requeues
            while True:
                exchange_rec = rec_handler.next_record()
                if exchange_rec is None:
                    break
                group = [exchange_rec]
                while len(group) < self.group_size:
                    more = rec_handler.try_next_record()
                    if more is None:
                        break
                    group.append(more)
                for exchange_rec, filtered_rec in zip(group, self._filter_group(group)):
                    rec_handler.send_record_to_outbound_exchange(exchange_rec.record_id, filtered_rec, exchange_rec.table_aliases)
                    rec_handler.complete_record(exchange_rec)
                    self.records_processed += 1
                    self.rows_in += exchange_rec.record.num_rows
                    self.rows_out += filtered_rec.num_rows
        finally:
            rec_handler.close()


def _schema_of(rec):
    if hasattr(rec, "schema"):
        return rec.schema
    return tuple(zip(rec.column_names, rec.column_formats))


class FilterTaskBuilder(TaskBuilder):
    """The GPU filter operator. Swap it in with `OperatorTaskRegistry.add_filter_task_builder` -- the planner's
    DAG and the exchanges are untouched (operator_task_registry.rs:51-57)."""

    def __init__(self, filter_fn=None, group_size: int = 1):
        self._filter_fn = filter_fn
        self._group_size = group_size

    def build(self, op_in_config, inbound_exchanges, outbound_exchange):
        task = FilterTask(op_in_config, FilterConfig.try_from(op_in_config), inbound_exchanges, outbound_exchange,
                          filter_fn=self._filter_fn, group_size=self._group_size)

        def run():
            try:
                task.async_main()
                return None
            except Exception as err:   # noqa: BLE001 -- any error ends the instance (producer_operator.rs:179-183)
                return err

        run.task = task
        return run


# ---- read_files (the table function in front of the path) -----------------------------------------------------
class ReadFilesTask:
    """read_files_task.rs:129-291 with `read_records` (:233-282) on the GPU: every matching Parquet file is opened through a
    RANGE reader (the footer, then exactly the chunks that are decoded -- the reference reads through opendal ranges), its row
    groups are decoded in HBM (`chq_parquet_read_columns`) and forwarded as device-resident records cut into zero-copy slices
    of `max_rows_per_batch` rows; `columns` prunes the scan (the reference's DEV_NOTES.md:123).  Files the decoder refuses
    (status 30: ZSTD, nested columns, ...) are read with pyarrow on the host -- the stand-in for the parquet crate."""

    def __init__(self, op_in_config: OperatorInstanceConfig, task: ReadFilesOperatorTask, storage_root: str, outbound_exchange,
                 ctx=None, columns: Optional[Sequence[str]] = None, device_records: bool = True):
        self.operator_instance_config = op_in_config
        self.config = task
        self.storage_root = storage_root
        self.outbound_exchange = outbound_exchange
        self._ctx = ctx
        self.columns = list(columns) if columns is not None else None
        self.device_records = device_records
        self.record_id = 0
        self.files_read: List[str] = []
        self.bytes_fetched = 0
        self.host_fallbacks = 0

    def _context(self):
        if self._ctx is None:
            self._ctx = record_utils.Context(self.operator_instance_config.device_id)
        return self._ctx

    def _send(self, rec_handler, record) -> None:
        aliases = record_utils.get_record_table_aliases(self.config.alias, record)
        rec_handler.send_record_to_outbound_exchange(self.record_id, record, aliases)
        self.record_id += 1

    def _read_records(self, rec_handler, path: str) -> None:
        size = os.path.getsize(path)
        step = max(1, int(self.config.max_rows_per_batch))
        with open(path, "rb") as fh:
            def read(offset: int, length: int) -> bytes:
                self.bytes_fetched += length
                return os.pread(fh.fileno(), length, offset)
            try:
                pf = record_utils.ParquetFile(None, reader=read, size=size)
            except record_utils.ChqError as err:
                if err.code != 30:
                    raise
                pf = None
            groups = None
            if pf is not None:
                try:
                    groups = pf.read_row_groups(ctx=self._context(), device_result=self.device_records, columns=self.columns)
                except record_utils.ChqError as err:
                    if err.code != 30:
                        raise
                finally:
                    pf.close()
        if groups is None:   # outside the GPU decoder's scope: the host reader, same batch size
            import pyarrow.parquet as pq
            self.host_fallbacks += 1
            for rec in pq.ParquetFile(path).iter_batches(batch_size=step, columns=self.columns):
                self._send(rec_handler, rec)
            return
        for group in groups:
            n = group.num_rows
            if n <= step:
                self._send(rec_handler, group)
                continue
            for at in range(0, n, step):
                self._send(rec_handler, group.slice(at, min(step, n - at)))

    def async_main(self) -> None:
        import glob
        rec_handler = RecordHandler.initiate(self.operator_instance_config, [], self.outbound_exchange)
        try:
            for path in sorted(glob.glob(os.path.join(self.storage_root, self.config.path.lstrip("/")), recursive=True)):
                self._read_records(rec_handler, path)
                self.files_read.append(path)
        finally:
            rec_handler.close()


class ReadFilesTaskBuilder(TaskBuilder):
    """The GPU read_files table function (`rust/gpu_read_files_task.rs` is the same task for the reference's registry)."""

    def __init__(self, storage_root: str, columns: Optional[Sequence[str]] = None, device_records: bool = True):
        self._root = storage_root
        self._columns = columns
        self._device_records = device_records

    def build(self, op_in_config, inbound_exchanges, outbound_exchange):
        if not isinstance(op_in_config.task, ReadFilesOperatorTask):
            raise ValueError("operator instance config is not a read_files task")
        task = ReadFilesTask(op_in_config, op_in_config.task, self._root, outbound_exchange, columns=self._columns,
                             device_records=self._device_records)

        def run():
            try:
                task.async_main()
                return None
            except Exception as err:   # noqa: BLE001
                return err

        run.task = task
        return run


# ---- materialize ----------------------------------------------------------------------------------------------
@dataclasses.dataclass(frozen=True)
class MaterializeFilesConfig:
    data_format: str
    fields: Sequence[A.SelectItem]

    @staticmethod
    def try_from(op_in_config: OperatorInstanceConfig) -> "MaterializeFilesConfig":
        if not isinstance(op_in_config.task, MaterializeFilesOperatorTask):
            raise ValueError("operator instance config is not a materialize files task")
        return MaterializeFilesConfig(op_in_config.task.data_format, op_in_config.task.fields)


class MaterializeFilesTask:
    def __init__(self, op_in_config, config: MaterializeFilesConfig, inbound_exchanges, outbound_exchange,
                 storage_root: str, project_fn=None, ctx=None):
        self.operator_instance_config = op_in_config
        self.materialize_file_config = config
        self.inbound_exchanges = inbound_exchanges
        self.outbound_exchange = outbound_exchange
        self.storage_root = storage_root
        self._project = project_fn
        self._ctx = ctx
        self.files_written: List[str] = []

    def _project_record(self, rec, aliases):
        if self._project is not None:
            return self._project(self.materialize_file_config.fields, rec, aliases)
        if self._ctx is None:
            self._ctx = record_utils.Context(self.operator_instance_config.device_id)
        # the projected batch stays in HBM: the Parquet pages are encoded there too
        return record_utils.project_record(self.materialize_file_config.fields, rec, aliases, ctx=self._ctx, device_result=True)

    def _write_parquet(self, proj_rec, path: str) -> None:
        """materialize_files_task.rs:128-141 (AsyncArrowWriter ... write ... close).  A device-resident result is encoded on
        the GPU (`chq_record_to_parquet`) and only the finished file image reaches the host; column types that encoder does
        not write (and host-resident results, e.g. from an injected `project_fn`) go through pyarrow's writer."""
        import pyarrow as pa
        import pyarrow.parquet as pq
        if isinstance(proj_rec, record_utils.DeviceRecordBatch):
            try:
                image = record_utils.record_to_parquet(proj_rec, ctx=self._ctx, copy=False)
            except record_utils.ChqError as e:
                if e.code != 30:   # NotSupported: a type outside Int32/Int64/Float32/Float64/Boolean/Utf8
                    raise
                proj_rec = proj_rec.to_host()
            else:
                try:
                    with open(path, "wb") as f:
                        f.write(image.view)     # straight out of the library's host buffer
                finally:
                    image.release()
                return
        pq.write_table(pa.Table.from_batches([proj_rec]), path)

    def async_main(self) -> None:
        """materialize_files_task.rs:68-170: pull -> project -> write /query_results/<uuid>/rec_<id>.parquet -> ack."""
        rec_handler = RecordHandler.initiate(self.operator_instance_config, self.inbound_exchanges, self.outbound_exchange)
        query_uuid = uuid.UUID(int=self.operator_instance_config.query_id)
        out_dir = os.path.join(self.storage_root, "query_results", str(query_uuid))
        os.makedirs(out_dir, exist_ok=True)
        try:
            while True:
                exchange_rec = rec_handler.next_record()
                if exchange_rec is None:
                    break
                proj_rec = self._project_record(exchange_rec.record, exchange_rec.table_aliases)
                path = os.path.join(out_dir, f"rec_{exchange_rec.record_id}.parquet")
                self._write_parquet(proj_rec, path)
                self.files_written.append(path)
                rec_handler.complete_record(exchange_rec)
        finally:
            rec_handler.close()


class MaterializeFilesTaskBuilder(TaskBuilder):
    def __init__(self, storage_root: str, project_fn=None):
        self.storage_root = storage_root
        self._project_fn = project_fn

    def build(self, op_in_config, inbound_exchanges, outbound_exchange):
        task = MaterializeFilesTask(op_in_config, MaterializeFilesConfig.try_from(op_in_config), inbound_exchanges,
                                    outbound_exchange, self.storage_root, project_fn=self._project_fn)

        def run():
            try:
                task.async_main()
                return None
            except Exception as err:   # noqa: BLE001
                return err

        run.task = task
        return run


def build_default_operator_task_registry(storage_root: str) -> OperatorTaskRegistry:
    """operator_task_registry.rs:150-162 with the GPU builders swapped in."""
    return (OperatorTaskRegistry()
            .add_table_func_task_builder("read_files", ReadFilesTaskBuilder(storage_root))
            .add_filter_task_builder(FilterTaskBuilder())
            .add_materialize_files_builder(MaterializeFilesTaskBuilder(storage_root), ["parquet"]))
