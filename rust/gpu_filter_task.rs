//! `GpuFilterTaskBuilder`: the reference's FilterTaskBuilder (operators/filter_tasks/filter_task.rs:152-199) with
//! the one compute call swapped for the GPU path.  Everything else -- RecordHandler protocol, message consumer,
//! restricted task tracker, oneshot completion -- is the reference's own code and stays unchanged, which is what
//! makes this a drop-in: register it with
//!     OperatorTaskRegistry::new().add_filter_task_builder(Box::new(GpuFilterTaskBuilder { device_id }))
//! in place of `FilterTaskBuilder::new()` (operators/operator_task_registry.rs:150-162).
//! Source only (no Rust toolchain in the build image).
//!
//! The only lines that differ from filter_task.rs are marked `// GPU`.

// ... same `use` list as filter_task.rs, plus:
use super::gpu_record_utils::{self, GpuContext};

impl FilterTask {
    async fn async_main(&mut self, ct: CancellationToken) -> Result<()> {
        let gpu = GpuContext::new(self.device_id)?;                                            // GPU: one context per instance
        let mut rec_handler = exchange_handlers::record_handler::RecordHandler::initiate(
            ct.child_token(), &self.operator_instance_config, &mut self.operator_pipe,
            self.msg_reg.clone(), self.msg_router_state.clone()).await?;
        loop {
            let exchange_rec = rec_handler.next_record(ct.child_token(), &mut self.operator_pipe, None).await?;
            match exchange_rec {
                Some(exchange_rec) => {
                    let filtered_rec = gpu_record_utils::filter_record(                        // GPU: was record_utils::filter_record
                        &gpu, exchange_rec.record.clone(), &exchange_rec.table_aliases, &self.filter_config.expr)?;
                    rec_handler.send_record_to_outbound_exchange(
                        &mut self.operator_pipe, exchange_rec.record_id.clone(), filtered_rec,
                        exchange_rec.table_aliases.clone()).await?;
                    rec_handler.complete_record(&mut self.operator_pipe, exchange_rec).await?;
                }
                None => break,
            }
        }
        if let Err(err) = rec_handler.close().await { error!("{}", err); }
        Ok(())
    }
}
