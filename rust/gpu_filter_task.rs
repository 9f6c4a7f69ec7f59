//! `GpuFilterTaskBuilder` -- the filter operator task of ChapterhouseDB with its per-batch compute on an MI355X.
//!
//! Drop-in for `FilterTaskBuilder` (operators/filter_tasks/filter_task.rs:152-199 of the reference): it implements the
//! same plugin trait (`TaskBuilder`, operators/traits.rs:22-36), parses the same config (`FilterConfig::try_from`,
//! filter_tasks/conversions.rs:16-36), spawns exactly one future on the restricted tracker, reports through the same
//! oneshot and hands back a `MessageConsumer` that whitelists the same replies (filter_task.rs:209-261).  Swap it in
//! where the default registry is built (operators/operator_task_registry.rs:150-162):
//!
//! ```ignore
//! OperatorTaskRegistry::new()
//!     .add_table_func_task_builder(Box::new(ReadFilesTaskBuilder::new()), Box::new(ReadFilesSyntaxValidator::new()))?
//!     .add_filter_task_builder(Box::new(GpuFilterTaskBuilder::new(GpuPlacement::ByInstanceId, 64)))?
//!     .add_materialize_files_builder(Box::new(GpuMaterializeFilesTaskBuilder::new(GpuPlacement::ByInstanceId)), vec![DataFormat::Parquet])?
//! ```
//!
//! What differs from the reference task is the body of the loop (filter_task.rs:86-125): instead of one
//! `record_utils::filter_record` call per 10 000-row batch, the task drains up to `group_size` records that are queued
//! right now and filters them with ONE kernel launch (`gpu_record_utils::filter_records` = `chq_filter_records`).
//! Each output keeps its input record id and is sent / acked individually, in order, so the exchange protocol
//! (send -> ack -> complete, filter_task.rs:106-118) and the per-record heartbeats are unchanged.
//!
//! Source only: the build image has no Rust toolchain (INTEGRATION.md section 1 lists what to add to Cargo.toml).
//! Place under src/handlers/operator_handler/operators/gpu_tasks/ next to gpu_record_utils.rs and chq_sys.rs.
use std::sync::Arc;
use std::time::Duration;

use anyhow::{Error, Result};
use tokio::sync::{oneshot, Mutex};
use tokio_util::sync::CancellationToken;
use tracing::{debug, error};

use crate::handlers::exchange_handlers::record_handler::{ExchangeRecord, RecordHandler};
use crate::handlers::message_handler::messages::{
    self,
    message::{Message, MessageName},
};
use crate::handlers::message_handler::{MessageRegistry, Pipe};
use crate::handlers::message_router_handler::{MessageConsumer, MessageRouterState};
use crate::handlers::operator_handler::operator_handler_state::OperatorInstanceConfig;
use crate::handlers::operator_handler::operators::filter_tasks::FilterConfig;
use crate::handlers::operator_handler::operators::operator_task_trackers::RestrictedOperatorTaskTracker;
use crate::handlers::operator_handler::operators::traits::TaskBuilder;
use crate::handlers::operator_handler::operators::ConnectionRegistry;

use super::gpu_record_utils::{self, GpuContext};

/// Which GPU an operator instance computes on.  The scheduler knows nothing about devices (worker config carries only
/// `compute{instances, memory_in_mib, cpu_in_thousandths}`), so the builder decides.
#[derive(Debug, Clone, Copy)]
pub enum GpuPlacement {
    /// every instance on this device
    Fixed(i32),
    /// instance id modulo the visible device count: one filter instance per GPU when the query runs N instances
    ByInstanceId,
}

impl GpuPlacement {
    pub fn device_for(&self, op_in_config: &OperatorInstanceConfig) -> Result<i32> {
        match self {
            GpuPlacement::Fixed(d) => Ok(*d),
            GpuPlacement::ByInstanceId => {
                let n = gpu_record_utils::device_count()?;
                Ok((op_in_config.id % n as u128) as i32)
            }
        }
    }
}

#[derive(Debug, Clone)]
pub struct GpuFilterTaskBuilder {
    placement: GpuPlacement,
    /// records drained per kernel launch; 1 reproduces the reference's one-call-per-record loop
    group_size: usize,
}

impl GpuFilterTaskBuilder {
    pub fn new(placement: GpuPlacement, group_size: usize) -> GpuFilterTaskBuilder {
        GpuFilterTaskBuilder { placement, group_size: group_size.max(1) }
    }
}

impl TaskBuilder for GpuFilterTaskBuilder {
    fn build(
        &self,
        op_in_config: OperatorInstanceConfig,
        operator_pipe: Pipe,
        msg_reg: Arc<MessageRegistry>,
        _conn_reg: Arc<ConnectionRegistry>,
        message_router_state: Arc<Mutex<MessageRouterState>>,
        tt: &mut RestrictedOperatorTaskTracker,
        ct: CancellationToken,
    ) -> Result<(oneshot::Receiver<Option<Error>>, Box<dyn MessageConsumer>)> {
        let filter_config = FilterConfig::try_from(&op_in_config)?;
        let device_id = self.placement.device_for(&op_in_config)?;
        let consumer: Box<dyn MessageConsumer> = Box::new(GpuTaskConsumer { msg_reg: msg_reg.clone(), sends_records: true });
        let mut task = GpuFilterTask {
            operator_instance_config: op_in_config,
            filter_config,
            device_id,
            group_size: self.group_size,
            operator_pipe,
            msg_reg,
            msg_router_state: message_router_state,
        };
        let (done_tx, done_rx) = oneshot::channel();
        // at most one spawned future per producer operator (operator_task_trackers.rs:28-40)
        tt.spawn(async move {
            let outcome = task.run(ct).await.err();
            if let Some(err) = &outcome {
                error!("gpu filter task failed: {:?}", err);
            }
            if done_tx.send(outcome).is_err() {
                error!("gpu filter task: the producer operator dropped its completion receiver");
            }
        })?;
        Ok((done_rx, consumer))
    }
}

struct GpuFilterTask {
    operator_instance_config: OperatorInstanceConfig,
    filter_config: FilterConfig,
    device_id: i32,
    group_size: usize,
    operator_pipe: Pipe,
    msg_reg: Arc<MessageRegistry>,
    msg_router_state: Arc<Mutex<MessageRouterState>>,
}

impl GpuFilterTask {
    async fn run(&mut self, ct: CancellationToken) -> Result<()> {
        // One library context (HIP stream, scratch, staging buffers) per operator instance: the library is re-entrant
        // across contexts and the reference runs one batch at a time per instance (filter_task.rs:86-125).
        let gpu = GpuContext::new(self.device_id)?;
        let mut records = RecordHandler::initiate(
            ct.child_token(),
            &self.operator_instance_config,
            &mut self.operator_pipe,
            self.msg_reg.clone(),
            self.msg_router_state.clone(),
        )
        .await?;
        debug!(operator_instance_id = self.operator_instance_config.id, device_id = self.device_id, "gpu filter task started");

        let mut group: Vec<ExchangeRecord> = Vec::with_capacity(self.group_size);
        loop {
            // block for the first record of a group, then take whatever else is queued right now
            match records.next_record(ct.child_token(), &mut self.operator_pipe, None).await? {
                Some(first) => group.push(first),
                None => break, // NoneLeft: the inbound exchange is drained
            }
            while group.len() < self.group_size {
                let more = records
                    .next_record(ct.child_token(), &mut self.operator_pipe, Some(chrono::Duration::zero()))
                    .await;
                match more {
                    Ok(Some(rec)) => group.push(rec),
                    // timeout (nothing queued) or NoneLeft: launch with what we have; NoneLeft shows up again on the
                    // next blocking pull
                    _ => break,
                }
            }

            let filtered = self.filter_group(&gpu, &group)?;
            for (exchange_rec, filtered_rec) in group.drain(..).zip(filtered.into_iter()) {
                // same record id out as in (filter_task.rs:109); ack the inbound exchange only after the outbound
                // exchange has the result (filter_task.rs:106-118)
                records
                    .send_record_to_outbound_exchange(
                        &mut self.operator_pipe,
                        exchange_rec.record_id,
                        filtered_rec,
                        exchange_rec.table_aliases.clone(),
                    )
                    .await?;
                records.complete_record(&mut self.operator_pipe, exchange_rec).await?;
            }
        }

        if let Err(err) = records.close().await {
            error!("{}", err);
        }
        debug!(operator_instance_id = self.operator_instance_config.id, "gpu filter task closed");
        Ok(())
    }

    /// One `chq_filter_records` call when the drained records share schema and aliases (they do unless two table
    /// functions feed one filter); otherwise record by record, which is the reference's loop.
    fn filter_group(&self, gpu: &GpuContext, group: &[ExchangeRecord]) -> Result<Vec<arrow::array::RecordBatch>> {
        let first = &group[0];
        let uniform = group
            .iter()
            .all(|g| g.table_aliases == first.table_aliases && g.record.schema() == first.record.schema());
        if group.len() > 1 && uniform {
            let recs: Vec<_> = group.iter().map(|g| g.record.clone()).collect();
            return gpu_record_utils::filter_records(gpu, &recs, &first.table_aliases, &self.filter_config.expr);
        }
        group
            .iter()
            .map(|g| gpu_record_utils::filter_record(gpu, g.record.clone(), &g.table_aliases, &self.filter_config.expr))
            .collect()
    }
}

// Give Duration a use even when chrono's re-export changes: the zero-wait pull above is the only timing knob.
#[allow(dead_code)]
const NO_WAIT: Duration = Duration::from_millis(0);

/// The replies a GPU producer task wants routed to its pipe -- the same whitelist as FilterConsumer
/// (filter_task.rs:209-261) and MaterializeFilesConsumer; shared by both GPU tasks.
#[derive(Debug, Clone)]
pub struct GpuTaskConsumer {
    pub msg_reg: Arc<MessageRegistry>,
    /// filter sends records on (wants SendRecordResponse); materialize does not
    pub sends_records: bool,
}

impl MessageConsumer for GpuTaskConsumer {
    fn consumes_message(&self, msg: &Message) -> bool {
        use messages::exchange::ExchangeRequests as Ex;
        use messages::query::QueryHandlerRequests as Qh;
        let wanted = match msg.msg.msg_name() {
            MessageName::Ping => self
                .msg_reg
                .try_cast_msg::<messages::common::Ping>(msg)
                .map(|m| matches!(m, messages::common::Ping::Pong)),
            MessageName::QueryHandlerRequests => self
                .msg_reg
                .try_cast_msg::<Qh>(msg)
                .map(|m| matches!(m, Qh::ListOperatorInstancesResponse { .. })),
            MessageName::ExchangeRequests => self.msg_reg.try_cast_msg::<Ex>(msg).map(|m| match m {
                Ex::GetNextRecordResponseRecord { .. }
                | Ex::GetNextRecordResponseNoneLeft
                | Ex::GetNextRecordResponseNoneAvailable
                | Ex::OperatorCompletedRecordProcessingResponse => true,
                Ex::SendRecordResponse { .. } => self.sends_records,
                _ => false,
            }),
            _ => return false,
        };
        wanted.unwrap_or_else(|err| {
            error!("{:?}", err);
            false
        })
    }
}
