//! `GpuMaterializeFilesTaskBuilder` -- the materialize operator task with `project_record` on an MI355X.
//!
//! Drop-in for `MaterializeFilesTaskBuilder` (operators/materialize_tasks/materialize_files_task.rs:177-230 of the
//! reference): same plugin trait (operators/traits.rs:22-36), same config (`MaterializeFilesConfig::try_from`), one
//! spawned future, same oneshot, same result location `/query_results/<query uuid>/rec_<record id>.parquet` on the
//! "default" connection (materialize_files_task.rs:83, 117-119), same reply whitelist.  Registered through
//! `add_materialize_files_builder(Box::new(GpuMaterializeFilesTaskBuilder::new(..)), vec![DataFormat::Parquet])`
//! (operators/operator_task_registry.rs:59-72).
//!
//! What differs is one call: `record_utils::project_record` (materialize_files_task.rs:110-114) becomes
//! `gpu_record_utils::project_record` (`chq_project_record`): one interpreted program per select item inside one HIP
//! kernel, pass-through columns copied host to host and never sent over PCIe.  The parquet encode and the opendal write
//! stay the reference's own code (out of the hot path).
//!
//! Source only: the build image has no Rust toolchain.
use std::path::PathBuf;
use std::sync::Arc;

use anyhow::{anyhow, Error, Result};
use tokio::sync::{oneshot, Mutex};
use tokio_util::sync::CancellationToken;
use tracing::{debug, error};
use uuid::Uuid;

use crate::handlers::exchange_handlers::record_handler::RecordHandler;
use crate::handlers::message_handler::{MessageRegistry, Pipe};
use crate::handlers::message_router_handler::{MessageConsumer, MessageRouterState};
use crate::handlers::operator_handler::operator_handler_state::OperatorInstanceConfig;
use crate::handlers::operator_handler::operators::materialize_tasks::MaterializeFilesConfig;
use crate::handlers::operator_handler::operators::operator_task_trackers::RestrictedOperatorTaskTracker;
use crate::handlers::operator_handler::operators::traits::TaskBuilder;
use crate::handlers::operator_handler::operators::ConnectionRegistry;

use super::gpu_filter_task::{GpuPlacement, GpuTaskConsumer};
use super::gpu_record_utils::{self, GpuContext};

#[derive(Debug, Clone)]
pub struct GpuMaterializeFilesTaskBuilder {
    placement: GpuPlacement,
}

impl GpuMaterializeFilesTaskBuilder {
    pub fn new(placement: GpuPlacement) -> GpuMaterializeFilesTaskBuilder {
        GpuMaterializeFilesTaskBuilder { placement }
    }
}

impl TaskBuilder for GpuMaterializeFilesTaskBuilder {
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
        let config = MaterializeFilesConfig::try_from(&op_in_config)?;
        let device_id = self.placement.device_for(&op_in_config)?;
        // materialize never forwards records (materialize_files_task.rs:95-153 has no send): no SendRecordResponse
        let consumer: Box<dyn MessageConsumer> = Box::new(GpuTaskConsumer { msg_reg: msg_reg.clone(), sends_records: false });
        let mut task = GpuMaterializeFilesTask {
            operator_instance_config: op_in_config,
            config,
            device_id,
            operator_pipe,
            msg_reg,
            conn_reg,
            msg_router_state: message_router_state,
        };
        let (done_tx, done_rx) = oneshot::channel();
        tt.spawn(async move {
            let outcome = task.run(ct).await.err();
            if let Some(err) = &outcome {
                error!("gpu materialize task failed: {:?}", err);
            }
            if done_tx.send(outcome).is_err() {
                error!("gpu materialize task: the producer operator dropped its completion receiver");
            }
        })?;
        Ok((done_rx, consumer))
    }
}

struct GpuMaterializeFilesTask {
    operator_instance_config: OperatorInstanceConfig,
    config: MaterializeFilesConfig,
    device_id: i32,
    operator_pipe: Pipe,
    msg_reg: Arc<MessageRegistry>,
    conn_reg: Arc<ConnectionRegistry>,
    msg_router_state: Arc<Mutex<MessageRouterState>>,
}

impl GpuMaterializeFilesTask {
    fn result_path(query_id: u128, record_id: u64) -> Result<String> {
        let mut path = PathBuf::from("/query_results");
        path.push(Uuid::from_u128(query_id).to_string());
        path.push(format!("rec_{}.parquet", record_id));
        path.to_str().map(|s| s.to_string()).ok_or_else(|| anyhow!("record path formatting returned None result"))
    }

    async fn run(&mut self, ct: CancellationToken) -> Result<()> {
        let gpu = GpuContext::new(self.device_id)?;
        let storage = self.conn_reg.get_operator("default")?;
        let mut records = RecordHandler::initiate(
            ct.child_token(),
            &self.operator_instance_config,
            &mut self.operator_pipe,
            self.msg_reg.clone(),
            self.msg_router_state.clone(),
        )
        .await?;
        debug!(operator_instance_id = self.operator_instance_config.id, device_id = self.device_id, "gpu materialize task started");

        while let Some(exchange_rec) = records.next_record(ct.child_token(), &mut self.operator_pipe, None).await? {
            let projected = gpu_record_utils::project_record(
                &gpu,
                &self.config.fields,
                exchange_rec.record.clone(),
                &exchange_rec.table_aliases,
            )?;

            // one parquet file per record id, written exactly as the reference does (materialize_files_task.rs:128-141)
            let path = Self::result_path(self.operator_instance_config.query_id, exchange_rec.record_id)?;
            let sink = storage.writer_with(&path).chunk(16 * 1024 * 1024).concurrent(4).await?;
            let mut parquet = parquet::arrow::AsyncArrowWriter::try_new(
                parquet_opendal::AsyncWriter::new(sink),
                projected.schema(),
                None,
            )?;
            parquet.write(&projected).await?;
            parquet.close().await?;

            records.complete_record(&mut self.operator_pipe, exchange_rec).await?;
        }

        if let Err(err) = records.close().await {
            error!("{}", err);
        }
        debug!(operator_instance_id = self.operator_instance_config.id, "gpu materialize task closed");
        Ok(())
    }
}
