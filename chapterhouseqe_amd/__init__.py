"""chapterhouseqe_amd -- MI355X-native filter / projection record kernels for ChapterhouseDB.

The hot path of alekLukanen/ChapterhouseQE's `filter` and `materialize` operators
(record_utils::{compute_value, filter_record, project_record}) as hand-written gfx950 HIP kernels behind
the C ABI of include/chq.h, plus the host-side mirror of the reference's interface.
"""
from . import sqlast, sqlparse  # noqa: F401
from .record_utils import (ChqError, Context, DeviceRecordBatch, compute_value, default_context,  # noqa: F401
                           filter_project_record, filter_record, filter_records, filter_records_coalesced,
                           get_record_table_aliases,
                           ipc_describe, plan_describe, project_record, record_from_ipc, record_to_ipc, RecordGroup,
                           ParquetFile, scan_parquet, record_to_parquet, records_to_parquet)

__all__ = ["sqlast", "sqlparse", "ChqError", "Context", "DeviceRecordBatch", "compute_value", "default_context",
           "filter_project_record", "filter_record", "filter_records", "filter_records_coalesced", "get_record_table_aliases", "plan_describe", "project_record",
           "RecordGroup", "ipc_describe", "record_from_ipc", "record_to_ipc", "ParquetFile", "scan_parquet", "record_to_parquet", "records_to_parquet"]
