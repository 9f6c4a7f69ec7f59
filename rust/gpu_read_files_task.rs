//! `GpuReadFilesTaskBuilder` -- the `read_files` table function with the Parquet page decode on an MI355X.
//!
//! Drop-in for `ReadFilesTaskBuilder` (operators/table_func_tasks/read_files_task.rs:293-338 of the reference): same plugin
//! trait (operators/traits.rs:22-36), same `TableFuncConfig` / `ReadFilesConfig` parsing and path globbing (the reference's
//! own `ReadFilesConfig` is reused, it would only need `pub(crate)` on `parse_config` / `parse_path_prefix` and its fields),
//! same outbound protocol (`RecordHandler::send_record_to_outbound_exchange`), same reply whitelist.  Registered through
//! `add_table_func_task_builder(Box::new(ReadFilesSyntaxValidator::new()), Box::new(GpuReadFilesTaskBuilder::new(..)))`
//! (operators/operator_task_registry.rs:35-49).
//!
//! What differs is `read_records` (read_files_task.rs:233-282): instead of `ParquetRecordBatchStreamBuilder` decoding on the
//! CPU, the file is opened through `chq_parquet_open_reader` with a RANGE callback over the same opendal operator (the footer,
//! then exactly the column chunks that are decoded), every row group is decoded in HBM by `chq_parquet_read_columns`
//! (UNCOMPRESSED and SNAPPY pages) and comes back as ONE batch, which is cut into zero-copy slices
//! of `max_rows_per_batch` rows so that downstream operators see the batch size the planner chose
//! (physical_planner.rs:323).  A file the GPU decoder refuses (`CHQ_ERR_NOT_SUPPORTED`: ZSTD, nested columns, DELTA
//! encodings) falls back to the reference's own reader.
//!
//! Source only: the build image has no Rust toolchain.
use std::ffi::{c_void, CStr};
use std::os::raw::{c_char, c_int};
use std::sync::Arc;

use anyhow::{anyhow, Error, Result};
use arrow::array::{RecordBatch, StructArray};
use arrow::ffi::{from_ffi, FFI_ArrowSchema};
use futures::StreamExt;
use tokio::sync::{oneshot, Mutex};
use tokio_util::sync::CancellationToken;
use tracing::{debug, error, warn};

use crate::handlers::exchange_handlers::record_handler::RecordHandler;
use crate::handlers::message_handler::{MessageRegistry, Pipe};
use crate::handlers::message_router_handler::{MessageConsumer, MessageRouterState};
use crate::handlers::operator_handler::operator_handler_state::OperatorInstanceConfig;
use crate::handlers::operator_handler::operators::operator_task_trackers::RestrictedOperatorTaskTracker;
use crate::handlers::operator_handler::operators::table_func_tasks::{ReadFilesConfig, TableFuncConfig};
use crate::handlers::operator_handler::operators::traits::TaskBuilder;
use crate::handlers::operator_handler::operators::{record_utils, ConnectionRegistry};

use super::chq_sys::{self, ArrowDeviceArray, ARROW_DEVICE_CPU};
use super::gpu_filter_task::{GpuPlacement, GpuTaskConsumer};
use super::gpu_record_utils::GpuContext;

const CHQ_ERR_NOT_SUPPORTED: c_int = 30;
/// row groups decoded per library call: their uploads and page decodes overlap inside one call
const ROW_GROUPS_PER_CALL: i32 = 16;

#[derive(Debug, Clone)]
pub struct GpuReadFilesTaskBuilder {
    placement: GpuPlacement,
}

impl GpuReadFilesTaskBuilder {
    pub fn new(placement: GpuPlacement) -> GpuReadFilesTaskBuilder {
        GpuReadFilesTaskBuilder { placement }
    }
}

impl TaskBuilder for GpuReadFilesTaskBuilder {
    fn build(
        &self,
        op_in_config: OperatorInstanceConfig,
        operator_pipe: Pipe,
        msg_reg: Arc<MessageRegistry>,
        conn_reg: Arc<ConnectionRegistry>,
        message_router_state: Arc<Mutex<MessageRouterState>>,
        tt: &mut RestrictedOperatorTaskTracker,
        ct: CancellationToken,
    ) -> Result<(oneshot::Receiver<Option<Error>>, Box<dyn MessageConsumer>)> {
        let table_func_config = TableFuncConfig::try_from(&op_in_config)?;
        let config = ReadFilesConfig::parse_config(&table_func_config)?;
        let device_id = self.placement.device_for(&op_in_config)?;
        // a producer: it sends records and waits for their SendRecordResponse (read_files_task.rs:355-401)
        let consumer: Box<dyn MessageConsumer> = Box::new(GpuTaskConsumer { msg_reg: msg_reg.clone(), sends_records: true });
        let mut task = GpuReadFilesTask {
            operator_instance_config: op_in_config,
            config,
            device_id,
            operator_pipe,
            msg_reg,
            conn_reg,
            msg_router_state: message_router_state,
            record_id: 0,
        };
        let (done_tx, done_rx) = oneshot::channel();
        tt.spawn(async move {
            let outcome = task.run(ct).await.err();
            if let Some(err) = &outcome {
                error!("gpu read_files task failed: {:?}", err);
            }
            if done_tx.send(outcome).is_err() {
                error!("gpu read_files task: the producer operator dropped its completion receiver");
            }
        })?;
        Ok((done_rx, consumer))
    }
}

struct GpuReadFilesTask {
    operator_instance_config: OperatorInstanceConfig,
    config: ReadFilesConfig,
    device_id: i32,
    operator_pipe: Pipe,
    msg_reg: Arc<MessageRegistry>,
    conn_reg: Arc<ConnectionRegistry>,
    msg_router_state: Arc<Mutex<MessageRouterState>>,
    record_id: u64,
}

/// What the range callback needs: a blocking view of the connection and the object's path.
struct RangeSource {
    op: opendal::BlockingOperator,
    path: String,
}

/// `chq_read_range_fn`: fill dst[0 .. length) with the object's bytes [offset, offset + length); 0 = ok.  Called from the
/// thread that runs the library call (a `spawn_blocking` thread below), never from the async executor.
unsafe extern "C" fn read_range(user: *mut c_void, offset: i64, length: i64, dst: *mut u8) -> c_int {
    let src = &*(user as *const RangeSource);
    match src.op.read_with(&src.path).range(offset as u64..(offset + length) as u64).call() {
        Ok(buf) if buf.len() as i64 == length => {
            std::ptr::copy_nonoverlapping(buf.to_bytes().as_ptr(), dst, length as usize);
            0
        }
        _ => 1,
    }
}

/// Every row group of one file as host `RecordBatch`es (decoded on the GPU), or `Ok(None)` when the decoder refuses the file.
fn decode_file(gpu: &GpuContext, src: &RangeSource, content_len: u64) -> Result<Option<Vec<RecordBatch>>> {
    let mut pq: *mut chq_sys::chq_parquet = std::ptr::null_mut();
    let mut err = [0 as c_char; 512];
    let rc = unsafe {
        chq_sys::chq_parquet_open_reader(content_len as i64, Some(read_range), src as *const RangeSource as *mut c_void, &mut pq, err.as_mut_ptr(), err.len())
    };
    if rc != 0 {
        let text = unsafe { CStr::from_ptr(err.as_ptr()) }.to_string_lossy().into_owned();
        return if rc == CHQ_ERR_NOT_SUPPORTED { Ok(None) } else { Err(anyhow!("[chq status {rc}] {text}")) };
    }
    struct Close(*mut chq_sys::chq_parquet);
    impl Drop for Close {
        fn drop(&mut self) {
            unsafe { chq_sys::chq_parquet_close(self.0) }
        }
    }
    let _close = Close(pq);
    let n_groups = unsafe { chq_sys::chq_parquet_num_row_groups(pq) };
    let mut out = Vec::with_capacity(n_groups as usize);
    let mut first = 0;
    while first < n_groups {
        let count = ROW_GROUPS_PER_CALL.min(n_groups - first);
        let mut arrs: Vec<ArrowDeviceArray> = (0..count).map(|_| ArrowDeviceArray::empty()).collect();
        let mut schemas: Vec<FFI_ArrowSchema> = (0..count).map(|_| FFI_ArrowSchema::empty()).collect();
        // every column, in file order (a planner that knows the query's columns passes their indices instead of null)
        let rc = unsafe {
            chq_sys::chq_parquet_read_columns(gpu.raw(), pq, first, count, std::ptr::null(), -1, ARROW_DEVICE_CPU, arrs.as_mut_ptr(), schemas.as_mut_ptr())
        };
        if rc == CHQ_ERR_NOT_SUPPORTED {
            return Ok(None);
        }
        gpu.check(rc)?;
        for (arr, schema) in arrs.into_iter().zip(schemas.into_iter()) {
            let data = unsafe { from_ffi(arr.array, &schema)? };
            out.push(RecordBatch::from(StructArray::from(data)));
        }
        first += count;
    }
    Ok(Some(out))
}

impl GpuReadFilesTask {
    async fn run(&mut self, ct: CancellationToken) -> Result<()> {
        debug!(
            operator_id = self.operator_instance_config.operator.id,
            operator_instance_id = self.operator_instance_config.id,
            device_id = self.device_id,
            "started gpu read_files task",
        );
        let mut rec_handler = RecordHandler::initiate(
            ct.child_token(),
            &self.operator_instance_config,
            &mut self.operator_pipe,
            self.msg_reg.clone(),
            self.msg_router_state.clone(),
        )
        .await?;
        let conn = self.conn_reg.get_operator(self.config.connection.as_deref().unwrap_or("default"))?;
        let gpu = Arc::new(std::sync::Mutex::new(GpuContext::new(self.device_id)?));
        let matcher = globset::Glob::new(self.config.path.as_str())?.compile_matcher();
        let mut lister = conn.lister_with(self.config.parse_path_prefix()).recursive(true).await?;
        loop {
            tokio::select! {
                entry = lister.next() => {
                    let entry = match entry { Some(e) => e?, None => break };
                    if matcher.is_match(entry.path()) {
                        self.read_file(&ct, entry.path(), &conn, &gpu, &mut rec_handler).await?;
                    }
                },
                _ = ct.cancelled() => break,
            }
        }
        if let Err(err) = rec_handler.close().await {
            error!("{}", err);
        }
        debug!(operator_instance_id = self.operator_instance_config.id, "closed gpu read_files task");
        Ok(())
    }

    async fn read_file(
        &mut self,
        ct: &CancellationToken,
        path: &str,
        conn: &opendal::Operator,
        gpu: &Arc<std::sync::Mutex<GpuContext>>,
        rec_handler: &mut RecordHandler,
    ) -> Result<()> {
        let content_len = conn.stat(path).await?.content_length();
        let src = RangeSource { op: conn.blocking(), path: path.to_string() };
        let gpu_for_call = gpu.clone();
        // the library call blocks (range reads, uploads, kernels): off the async executor
        let decoded = tokio::task::spawn_blocking(move || {
            let gpu = gpu_for_call.lock().map_err(|_| anyhow!("gpu context poisoned"))?;
            decode_file(&gpu, &src, content_len)
        })
        .await??;
        let row_groups = match decoded {
            Some(groups) => groups,
            None => {
                warn!(path, "the GPU decoder does not cover this file: reading it with the parquet crate");
                return self.read_file_on_the_host(ct, path, conn, rec_handler).await;
            }
        };
        let step = self.config.max_rows_per_batch.max(1);
        for group in row_groups {
            let mut at = 0;
            while at < group.num_rows() {
                if ct.is_cancelled() {
                    return Err(anyhow!("cancelled"));
                }
                let record = group.slice(at, step.min(group.num_rows() - at));   // zero-copy: the row group's buffers are shared
                at += record.num_rows();
                self.send(record, rec_handler).await?;
            }
        }
        Ok(())
    }

    /// read_files_task.rs:233-282, unchanged: the fallback for files outside the GPU decoder's scope
    async fn read_file_on_the_host(&mut self, ct: &CancellationToken, path: &str, conn: &opendal::Operator, rec_handler: &mut RecordHandler) -> Result<()> {
        let reader = conn.reader_with(path).gap(512 * 1024).chunk(16 * 1024 * 1024).concurrent(4).await?;
        let content_len = conn.stat(path).await?.content_length();
        let parquet_reader = parquet_opendal::AsyncReader::new(reader, content_len).with_prefetch_footer_size(512 * 1024);
        let mut stream = parquet::arrow::ParquetRecordBatchStreamBuilder::new(parquet_reader)
            .await?
            .with_batch_size(self.config.max_rows_per_batch)
            .build()?;
        while let Some(record) = stream.next().await {
            if ct.is_cancelled() {
                return Err(anyhow!("cancelled"));
            }
            self.send(record?, rec_handler).await?;
        }
        Ok(())
    }

    async fn send(&mut self, record: RecordBatch, rec_handler: &mut RecordHandler) -> Result<()> {
        let record_id = self.record_id;
        self.record_id += 1;
        let table_aliases = record_utils::get_record_table_aliases(&self.operator_instance_config.operator.operator_type, &record)?;
        rec_handler
            .send_record_to_outbound_exchange(&mut self.operator_pipe, record_id, record, table_aliases)
            .await
    }
}
