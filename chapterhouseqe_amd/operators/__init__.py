"""Operator / exchange layer around the record kernels (mirror of the reference's operator plugin API)."""
from .exchange_operator import NONE_AVAILABLE, NONE_LEFT, ExchangeOperator, RecordPool, RecordPoolError  # noqa: F401
from .record_handler import ExchangeRecord, RecordHandler, RecordHandlerError  # noqa: F401
from .tasks import (FilterConfig, FilterOperatorTask, FilterTask, FilterTaskBuilder, MaterializeFilesConfig,  # noqa: F401
                    MaterializeFilesOperatorTask, MaterializeFilesTask, MaterializeFilesTaskBuilder,
                    OperatorInstanceConfig, OperatorTaskRegistry, OperatorTaskRegistryError, TaskBuilder,
                    build_default_operator_task_registry)
