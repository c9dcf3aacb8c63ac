// vc_pass.cpp -- one Levenberg-Marquardt pass of the device pipeline as a fixed sequence of launches over two streams:
// which kernel goes where, the cross-stream hand-overs (device flags or events), the all-reduces of a sharded pass, graph capture.
#include "vc_calibrator.hpp"

int vc_calibrator::launch_pass_graph() {
  const bool flips = dv.imu_on && dv.weights_on;
  const int par = flips ? wcur : 0;
  if (!pass_graph[par]) {
    hipGraph_t g = nullptr;
    const int w0 = wcur;
    if (hipStreamBeginCapture(stream, hipStreamCaptureModeThreadLocal) != hipSuccess) { use_graphs = false; return enqueue_pass(false); }
    const int rc = enqueue_pass(false);
    const hipError_t e = hipStreamEndCapture(stream, &g);
    wcur = w0;
    if (rc != VC_OK || e != hipSuccess || !g || hipGraphInstantiate(&pass_graph[par], g, nullptr, nullptr, 0) != hipSuccess) {
      if (g) (void)hipGraphDestroy(g);
      pass_graph[par] = nullptr; use_graphs = false;
      (void)hipGetLastError();
      return enqueue_pass(false);
    }
    (void)hipGraphDestroy(g);
  }
  HIP_OK(hipGraphLaunch(pass_graph[par], stream));
  if (flips) wcur = 1 - wcur;
  return VC_OK;
}

int vc_calibrator::enqueue_pass(bool first_pass, bool events_only) {
  RoctxRange rr(first_pass ? "vicalib_amd: LM pass (first of a solve: + linearisation)" : "vicalib_amd: LM pass");
  const int D = dv.D;
  dv.merged = 0; dv.par = 0; dv.ctrl = d_ctrl.p; dv.ctrl_prev = d_ctrl.p + 1; dv.tail_deferred = 0; dv.hadd_early = 0; dv.part_ride = 0;
  if (dv.imu_on) {
    // UpdateImuWeights of the iteration callback (vicalibrator.h:691): linearise with the current weights, evaluate the
    // trial point with the updated ones.  The pass is a small graph over two streams: the weight update (which only needs the
    // accepted state and writes the other weight buffer) and the interval / block deltas of the trial point (which only need
    // the trial IMU parameters) run on the second, low-priority stream next to the chain solve -- every one of these kernels
    // is a few hundred latency-bound wavefronts, far from filling the chip on its own.  Everything on the critical path --
    // chain, both trial sweeps, decision -- stays on the main stream: kernels of one stream follow each other without a gap,
    // an event hand-over costs 5-13 us.
    // (the first pass of a solve: UpdateImuWeights() has just run on this very state (solve_once) -- the pass's own update would write the
    //  same numbers into the other buffer, 30 us on the second stream beside the first linearisation and the bottom chain level: the pass
    //  evaluates its trial point with the buffer it linearises with, and the buffers do not swap)
    const bool upd = dv.weights_on != 0 && !(first_pass && pre_weights_fresh && !events_only);      // (a stand-alone pass always updates)
    if (first_pass) pre_weights_fresh = false;
    // (sharded solves: flags when every rank has a device of its own -- vc_set_shard_rccl with more than one rank, or
    //  VICALIB_AMD_SHARD_FLAG_SYNC=1; the one-GPU gloo tests keep the events: several processes' waiting kernels would burn each
    //  other's time slices.  A time-out is lossless there too: the mark travels with the step scalars' all-reduce, all ranks resume)
    const bool fs = flag_sync && !serial_weights && (!sharded() || shard_flag_sync) && !use_graphs && !events_only;      // (a captured pass has fixed arguments and needs the events to fork the capture)
    ++pass_seq;
    dv.pass_id = pass_seq;
    wr_ring[pass_seq & 15] = wcur;              // (what a resume after a flag time-out restores: the weight buffer this pass reads)
    dv.sync_flags = d_sync.p; dv.sync_seq = fs ? pass_seq : 0; dv.final_wait = 0; dv.block_wait = 0; dv.sync_bound = (pass_seq >= sync_bound_from_pass) ? sync_bound : 800000;
    const bool fs_trial = fs && jac_on_stream2 && dv.n_tiles > 0;      // (no tiles: no trial sweep to publish the back-substitution's end)
    dv.final_wait = fs_trial ? pass_seq : 0;      // (k_imu_jac(trial) and k_final both look at it)
    dv.block_wait = fs_trial ? pass_seq : 0;      // (k_imu_block(trial))
    // The Jacobian sweeps at the head of the pass only run when the control record asks for a linearisation: the first pass
    // of a solve.  Afterwards the trial point is evaluated by the same sweeps in trial mode (below), which leave the next
    // linearisation behind if the step is accepted; after a rejected step the old one is still in place.
    if (!serial_weights) {
      bool block_done = false;
      if (first_pass && pre_weights_pending) {
        // the weight update that precedes a solve (solve_once) is still running on the main stream: the block deltas need the IMU
        // parameters only, not the weights -- they start from the event recorded ahead of it (35 us less per solve at cfg3)
        HIP_OK(hipStreamWaitEvent(stream2, ev_pre, 0));
        KT2("k_imu_block", launch_imu_delta(dv, stream2, 0));
        block_done = true;
      }
      pre_weights_pending = false;
      if (fs && !first_pass && prev_pass_signals) launch_wait_flag(dv, 0, pass_seq - 1, stream2);      // the previous pass's k_final
      else {
        HIP_OK(hipEventRecord(ev_state, stream));
        // the vision linearisation of a first pass depends on nothing the second stream does: it goes out before that stream's
        // launches (five runtime calls: the main stream would sit idle behind the preceding weight update for as long as they take)
        if (first_pass) KT("k_reproj_jac", launch_reproj_jac(dv, stream, 0));
        HIP_OK(hipStreamWaitEvent(stream2, ev_state, 0));
      }
      if (first_pass) {
        if (!block_done) KT2("k_imu_block", launch_imu_delta(dv, stream2, 0));
        KT2("k_imu_jac", launch_imu_jac(dv, wcur, stream2, 0));
        HIP_OK(hipEventRecord(ev_imujac, stream2));       // ahead of the weight update: the chain does not read the weights
      }
      // flag hand-overs: the weight update (500 wavefronts that take a SIMD's whole register file each) starts behind the bottom level of
      // the chain elimination, whose two-sided form needs the chip to itself; it is not needed before k_imu_jac(trial)
      if (upd && fs && !first_pass && weights_behind_l0 && chain_forward_launches(dv) >= 2) launch_wait_flag(dv, 7, pass_seq, stream2);
      if (upd) KT2("k_imu_weights", launch_imu_weights(dv, wcur, stream2));
      if (first_pass) HIP_OK(hipStreamWaitEvent(stream, ev_imujac, 0));
    } else {
      if (upd) KT("k_imu_weights", launch_imu_weights(dv, wcur, stream));
      if (first_pass) {
        KT("k_reproj_jac", launch_reproj_jac(dv, stream, 0));
        KT("k_imu_block", launch_imu_delta(dv, stream, 0)); KT("k_imu_jac", launch_imu_jac(dv, wcur, stream, 0));
      }
    }
    // (round 6) camera blocks, IMU block and chunk costs of the reduced system as side jobs of the chain's upper-level launches
    static const bool hadd_env = [] { const char* e = std::getenv("VICALIB_AMD_HADD_EARLY"); return !(e && e[0] == '0'); }();
    dv.hadd_early = (hadd_env && chain_hadd_early(dv)) ? 1 : 0;      // (sharded passes too: every rank's record is its own frames' share, the all-reduce of Sbuf adds them)
    // ... and, with flag hand-overs, the chunk records' fixed-order sums in the top level's launch instead of a launch of their own (the Gram
    // chunks all in that launch: no partial record of the top level's frames behind it)
    static const bool ride_env = [] { const char* e = std::getenv("VICALIB_AMD_PART_RIDE"); return !(e && e[0] == '0'); }();
    dv.part_ride = (ride_env && fs && dv.hadd_early && !top_gram_launch && dv.part_ready) ? 1 : 0;
    if (!dv.fold_l0) KT("k_chain_init", launch_chain_init(dv, stream));      // (fold: the bottom level's launch assembles its frames itself)
    KT("k_chain_fwd", launch_chain_fwd(dv, stream));
    if (dv.gram_top_stride == 0) KT("k_chain_gram", launch_chain_gram(dv, stream));      // (early Gram: the sums ride in the top level's launch)
    else if (top_gram_launch) KT("k_chain_gram(top)", launch_chain_gram_top(dv, stream));
    if (!dv.part_ride) KT("k_part_sum", launch_part_sum(dv, stream));
    int rc = VC_OK;
    // (round 6) the reduced solve's tail rides in the back-substitution's launch where that is one launch of k_chain_back_path
    static const bool defer_env = [] { const char* e = std::getenv("VICALIB_AMD_DEFER_TAIL"); return !(e && e[0] == '0'); }();
    dv.tail_deferred = (defer_env && chain_back_is_path(dv)) ? 1 : 0;
    if (sharded()) {
      KT("k_reduced(assemble)", launch_reduced(dv, 1, stream));
      KT("allreduce(S)", rc = do_allreduce(dv.Sbuf, D * D + 3 * D + 2, 0)); if (rc) return rc;
      KT("k_reduced(solve)", launch_reduced(dv, 2, stream));
    } else {
      KT("k_reduced", launch_reduced(dv, 0, stream));
    }
    // the trial IMU parameters exist: the interval deltas of the trial point run on the second stream next to the chain's
    // back-substitution (they depend on no pose)
    if (!serial_weights) {
      if (fs) launch_wait_flag(dv, 1, pass_seq, stream2);      // this pass's k_reduced
      else {
        HIP_OK(hipEventRecord(ev_reduced, stream));
        HIP_OK(hipStreamWaitEvent(stream2, ev_reduced, 0));
      }
      KT2("k_imu_block(trial)", launch_imu_delta(dv, stream2, 1));
    }
    {
      // (a captured pass freezes its arguments, pass_id among them: from the second replay on the ready words of the fused
      //  back-substitution would already hold a number >= it and its consumers would not wait -- one launch per level there)
      long long* const ready = dv.cready;
      if (use_graphs) dv.cready = nullptr;
      KT("k_chain_back", launch_chain_solve_b(dv, stream));
      dv.cready = ready;
    }
    // trial point: both sweeps in trial mode on the main stream, the IMU blocks with the weights this pass has just updated
    // (second stream: weight update, then the deltas -- ev_weights covers both); the decision follows without another
    // cross-stream hop (each costs 6-13 us on the device's timeline)
    if (upd) wcur = 1 - wcur;
    if (!serial_weights && jac_on_stream2) {
      // the IMU blocks' final stage (needs the trial poses) beside the vision sweep: the second stream is already past its
      // deltas when the back-substitution ends
      // (flag hand-overs: k_imu_block(trial), the kernel before it on the second stream, has waited for the flag the first
      // workgroup of k_reproj_jac(trial) sets -- one thread, before the kernel ended.  Letting k_imu_jac's own workgroups wait
      // at their entry was tried: 250 workgroups each invalidating the L2 under the running vision sweep, both kernels 2.3x
      // slower; a waiting kernel of its own costs 5 us on this stream's queue)
      if (!fs_trial) {
        HIP_OK(hipEventRecord(ev_back, stream));
        HIP_OK(hipStreamWaitEvent(stream2, ev_back, 0));
      }
      KT2("k_imu_jac(trial)", launch_imu_jac(dv, wcur, stream2, 1));
      if (!fs_trial) HIP_OK(hipEventRecord(ev_weights, stream2));      // (flag hand-overs: k_final ends on the second count of k_imu_jac's workgroups)
      KT("k_reproj_jac(trial)", launch_reproj_jac(dv, stream, 1));
      if (!fs_trial) HIP_OK(hipStreamWaitEvent(stream, ev_weights, 0));      // (flag hand-overs: k_final waits for the second stream itself)
    } else {
      if (!serial_weights) HIP_OK(hipEventRecord(ev_weights, stream2));
      KT("k_reproj_jac(trial)", launch_reproj_jac(dv, stream, 1));
      if (!serial_weights) HIP_OK(hipStreamWaitEvent(stream, ev_weights, 0));
      else KT("k_imu_block(trial)", launch_imu_delta(dv, stream, 1));
      KT("k_imu_jac(trial)", launch_imu_jac(dv, wcur, stream, 1));
    }
    if (sharded()) {
      KT("k_final(reduce)", launch_final(dv, 1, stream));
      KT("allreduce(step scalars)", rc = do_allreduce(dv.gath, world * kNumScal, 0)); if (rc) return rc;
      KT("k_final(decide)", launch_final(dv, 2, stream));
    } else {
      KT("k_final", launch_final(dv, 0, stream));
    }
    prev_pass_signals = fs;
    return VC_OK;
  }
  dv.sync_seq = 0; dv.final_wait = 0; dv.block_wait = 0;
  // merged decision (single process): control records alternate, pass k judges pass k-1 at the head of k_frame_schur
  const bool merged = merged_enabled && !use_graphs;      // (a captured graph has fixed kernel arguments)
  dv.shard_src = sharded() ? 1 : 0;
  if (merged) {
    if (first_pass) kpass = 0;
    dv.merged = 1; dv.par = kpass & 1; dv.ctrl = d_ctrl.p + (kpass & 1); dv.ctrl_prev = d_ctrl.p + ((kpass + 1) & 1);
    ++kpass;
  }
  if (first_pass || !dv.fused) KT("k_reproj_jac", launch_reproj_jac(dv, stream));
  KT("k_frame_schur+k_part_sum", launch_frame_schur(dv, stream));
  int rc = VC_OK;
  if (sharded()) {
    launch_reduced(dv, 1, stream);
    rc = do_allreduce(dv.Sbuf, D * D + 3 * D + 2, 0); if (rc) return rc;
    launch_reduced(dv, 2, stream);
  } else {
    KT("k_reduced", launch_reduced(dv, 0, stream));
  }
  KT("k_trial", launch_trial(dv, stream));
  if (sharded()) {
    launch_final(dv, 1, stream);
    rc = do_allreduce(dv.gath, world * kNumScal, 0); if (rc) return rc;
    if (!merged) launch_final(dv, 2, stream);      // merged: the next pass's frame elimination combines the ranks and decides
  } else if (!merged) {
    KT("k_final", launch_final(dv, 0, stream));
  }
  return VC_OK;
}
