// Host runtime + C ABI (include/gsv_engine.h) of the MI355X garbling engine: ONE translation unit in seven files.
// Device memory, streams and events are plain HIP runtime calls; there is NO CPU execution path for
// garble/evaluate — without a HIP device gsv_engine_create fails with GSV_ERR_DEVICE.
//   engine_internal.hpp       error reporting, device allocation, the deferred-release gate, the objects behind the opaque handles
//   engine_abi_record.ipp     gsv_recorder_*, gsv_program_*, gsv_engine_*, gsv_labels_from_seed
//   engine_session.ipp        program sessions
//   engine_plan.ipp           gsv_plan_*: built-in builder (single / dual), plan files, background compilation, plan recorder
//   engine_plan_session.ipp   plan sessions: schedule + device tables, inputs, window launches
//   engine_drain.ipp          streaming garbler: drain pipeline, garble || evaluate, safe-schedule fallback
//   engine_evaluate.ipp       evaluation, read-back, CBC-MAC helpers
// (The parts share file-local helpers and are included in this order inside one extern "C" block; build.py's dependency scan covers them.)
#include "engine_internal.hpp"

extern "C" {

#include "engine_abi_record.ipp"
#include "engine_session.ipp"
#include "engine_plan.ipp"
#include "engine_plan_session.ipp"
#include "engine_drain.ipp"
#include "engine_evaluate.ipp"

}  // extern "C"
