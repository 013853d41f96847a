// The linearisation of one Gauss-Newton round at the pose (R, t): every (tree, range) unit of this workgroup, pass by
// pass — transform, correspondence reuse or descent, gate, e, J, weight, accumulation (mad_icp.cpp:59-103 under
// pipeline.cpp:180-183).  TEXTUALLY included by icp_round (one round per launch) and icp_persist (all rounds in one
// launch) so that both kernels run the same instructions in the same order; not a translation unit of its own.
// Names it expects in scope — launch constants: QPT, TRACE, RPT, S, L, hi, nslots, u_first, k_first, r_first, job, moving,
// matched, corr, cache_leaf, cache_margin, min_ball, rho, b_ratio, inv_min_ball, opt_lds_top, opt_stage_min, s_top, s_exit,
// s_td, gate_file, Lv, phys ("Ranges", kernels.hip.h); per round: round, reuse, gate_reuse, mark_matched, stage_hint, R[9], t[3],
// wear_alpha, wear_beta, pv0 / cmar0 / cword0 (the first pass's pose-independent loads, already issued); state it updates:
// desc_tree, staged_tree, acc[kAcc], visits,
// walked_visits, walked.  MADICP_TID: the thread index (threadIdx.x; icp_persist hands in a per-round copy the compiler
// cannot prove loop-invariant, so that per-lane addresses are recomputed every round instead of hoisted and spilled).
  int k = k_first, r = r_first;
  for (int u = u_first; u < hi; u += nslots, r += nslots) {
    while (r >= RPT) {  // (k, r) follow u without a division
      r -= RPT;
      ++k;
    }
    const int i_end = min(Lv, (r + 1) * S);  // (virtual indices: kernels.hip.h, "Ranges")
    if (k != desc_tree) {  // (workgroup-uniform; only workgroups with several units get here)
      __syncthreads();     // nobody still reads the previous descriptor
      if (MADICP_TID < 11)
        reinterpret_cast<long long*>(&s_td)[MADICP_TID] =
            ((const __attribute__((address_space(1))) long long*)(uintptr_t)&job->trees[k])[MADICP_TID];
      __syncthreads();
      desc_tree = k;
    }
    const TreeDesc& td = s_td;
    // staging costs ~2 x n_top lane-loads per workgroup: only worth it when the unit walks many leaves — and only
    // when somebody actually has to walk (with correspondence reuse most rounds need no walk at all)
    const int n_top_avail = (opt_lds_top && i_end - r * S >= opt_stage_min) ? min(td.n_top, kTopMax) : 0;
    // wear of a pair of this tree up to this round: A = |p| wear_alpha + wear_k (kernels.hip.h, "Bookkeeping without a store")
    const double wear_k = wear_beta + (double)round * (1e-11 * (td.rho + fabs(td.origin[0]) + fabs(td.origin[1]) + fabs(td.origin[2]) + 1.0));

    for (int base = r * S; base < i_end; base += QPT * kBlock) {
      double px[QPT], py[QPT], pz[QPT], pn[QPT], q0[QPT], q1[QPT], q2[QPT], margin[QPT];
      bool valid[QPT], walk[QPT];
      int leaf[QPT], depth[QPT];
      // every load of this pass that does not depend on another one is issued first — the leaf's coordinates and
      // its cached correspondence — so a walk-free pass is two memory round trips (these, then the leaf record)
      vd4 pv[QPT];
      float cmar[QPT];   // the pair's threshold on file (kernels.hip.h, "One threshold per pair"): |T| = how far it may wear, T < 0: rejected
      float tkeep[QPT];  // the leaf's own threshold (what the pair keeps when the gate lets it through)
      unsigned int cword[QPT];
      bool skip[QPT];  // gate reuse: the pair keeps its leaf and is still rejected — nothing to fetch, nothing to add
      double wearv[QPT];
#pragma unroll
      for (int j = 0; j < QPT; ++j) {
        const int v = base + j * kBlock + MADICP_TID;
        const int i = phys(r, v);
        valid[j] = v < i_end && i < L;
        pv[j] = vd4{0.0, 0.0, 0.0, 0.0};
        cmar[j] = 0.f;
        cword[j] = 0u;
        skip[j] = false;
        if (u == u_first && base == r * S) {  // (workgroup-uniform) already fetched before the solve prologue
          pv[j] = pv0[j];
          cmar[j] = cmar0[j];
          cword[j] = cword0[j];
        } else if (valid[j]) {
          pv[j] = ((gptr_d4)(uintptr_t)moving)[i];
          if (reuse) {
            const long long ci = (long long)k * L + i;
            cmar[j] = ((const __attribute__((address_space(1))) float*)(uintptr_t)cache_margin)[ci];
            cword[j] = ((const __attribute__((address_space(1))) unsigned int*)(uintptr_t)cache_leaf)[ci];
          }
        }
      }
#ifdef MADICP_STAMPS
      if (u == u_first && base == r * S) { MADICP_STAMP(7); }
      if (u == u_first && base == r * S + QPT * kBlock) { MADICP_STAMP_WAIT(); MADICP_STAMP(10); }
#endif
#pragma unroll
      for (int j = 0; j < QPT; ++j) {
        const int i = phys(r, base + j * kBlock + MADICP_TID);
        const vd4 p = pv[j];
        px[j] = p.x; py[j] = p.y; pz[j] = p.z; pn[j] = p.w;
        // ml = X * p  (Isometry3d * Vector3d: linear()*p + translation(), mad_icp.cpp:78)
#ifdef MADICP_XFORM_HOMOGENEOUS  // (the homogeneous-product order: oracle/linalg.h apply(); same operation count)
        q0[j] = ((R[0] * p.x + R[1] * p.y) + R[2] * p.z) + t[0];
        q1[j] = ((R[3] * p.x + R[4] * p.y) + R[5] * p.z) + t[1];
        q2[j] = ((R[6] * p.x + R[7] * p.y) + R[8] * p.z) + t[2];
#else
        q0[j] = t[0] + dots(R[0], R[1], R[2], p.x, p.y, p.z);
        q1[j] = t[1] + dots(R[3], R[4], R[5], p.x, p.y, p.z);
        q2[j] = t[2] + dots(R[6], R[7], R[8], p.x, p.y, p.z);
#endif
        walk[j] = valid[j];
        margin[j] = 3.0e38;
        leaf[j] = 0;
        depth[j] = 0;
        // everything this leaf can have moved against this tree since round 0, rounding included (an upper bound that only
        // grows: thresholds measured against it are never rewritten while they hold)
        const double wear = __builtin_fma(p.w, wear_alpha, wear_k);
        wearv[j] = wear;
        tkeep[j] = fabsf(cmar[j]);
        if (reuse && valid[j] && (double)tkeep[j] > wear) {  // every side test of the old path keeps its sign: same leaf, same depth
          leaf[j] = (int)(cword[j] & kCacheIdxMask);
          depth[j] = (int)(cword[j] >> 26);
          walk[j] = false;
        }
        // same leaf as when the slack was measured, and still further outside its ball than it can have moved since?  (a negative
        // threshold is min(the leaf's, the slack's): above the wear, both hold)
        skip[j] = gate_reuse && valid[j] && !walk[j] && cmar[j] < 0.f;
      }
      if (u == u_first && base == r * S) { MADICP_STAMP(3); }
      if (u == u_first && base == r * S + QPT * kBlock) { MADICP_STAMP(11); }
      {
#pragma unroll
        for (int j = 0; j < QPT; ++j) walked |= walk[j];
        if (n_top_avail > 0 && k != staged_tree && stage_hint) {  // (workgroup-uniform condition) copy the top levels into LDS
          if (staged_tree >= 0) __syncthreads();  // nobody may still be walking the previous tree's copy
          gptr_u4 gt = (gptr_u4)(uintptr_t)td.top;
          gptr_u4 ge = (gptr_u4)(uintptr_t)td.top_exit;
          for (int e = MADICP_TID; e < n_top_avail; e += kBlock) {
            s_top[e] = gt[e];
            reinterpret_cast<vu4*>(s_exit)[e] = ge[e];
          }
          __syncthreads();
          staged_tree = k;
        }
        const int n_top = (k == staged_tree) ? n_top_avail : 0;
        // The lane's QPT leaves share their LOADS (coordinates, cache, leaf record: issued together above and below),
        // but they are WALKED one after the other: interleaved walks make every step wait for the slowest of
        // 64*QPT lanes and were measured slower than back-to-back ones.
        int widx[QPT], wleaf[QPT], wdepth[QPT];
#pragma unroll
        for (int j = 0; j < QPT; ++j) {
          const double a0[1] = {q0[j]}, a1[1] = {q1[j]}, a2[1] = {q2[j]};
          const bool wv[1] = {walk[j]};
          int xi[1], xl[1], xd[1];
          double xm[1] = {margin[j]};
          bool any_walk = walk[j];
          if (QPT > 1) any_walk = __any(walk[j]);  // skip the whole (wave-uniform) call when nobody in the wave walks
          if (any_walk) {
            descend_multi<1>(td, s_top, s_exit, n_top, a0, a1, a2, wv, xi, xl, xd, xm);
            widx[j] = xi[0]; wleaf[j] = xl[0]; wdepth[j] = xd[0]; margin[j] = xm[0];
          } else {
            widx[j] = 0; wleaf[j] = 0; wdepth[j] = 0;
          }
        }
#pragma unroll
        for (int j = 0; j < QPT; ++j) {
          if (walk[j]) {
            leaf[j] = wleaf[j];
            depth[j] = wdepth[j];
            walked_visits += (unsigned int)wdepth[j];
            if (cache_leaf) {
              const long long ci = (long long)k * L + phys(r, base + j * kBlock + MADICP_TID);
              const bool cacheable = wdepth[j] <= kCacheMaxDepth && (unsigned int)wleaf[j] <= kCacheIdxMask;
              cache_leaf[ci] = (unsigned int)wleaf[j] | ((unsigned int)wdepth[j] << 26);
              tkeep[j] = cacheable ? __double2float_rd(margin[j] + wearv[j]) : 0.f;  // (filed below, once the gate has spoken)
            }
          }
        }
      }
#pragma unroll
      for (int j = 0; j < QPT; ++j)
        if (valid[j]) visits += (unsigned int)depth[j];
      if (u == u_first && base == r * S) { MADICP_STAMP(4); }

#pragma unroll
      for (int j = 0; j < QPT; ++j) {
        if (!valid[j]) continue;
        const int i = phys(r, base + j * kBlock + MADICP_TID);
        if (skip[j]) {  // still rejected (gate reuse): mad_icp.cpp:83 `continue`
          if (TRACE && corr) corr[(long long)td.slot * L + i] = static_cast<uint32_t>(leaf[j]) | 0x80000000u;
          continue;
        }
        // the matched leaf's record: one 64-byte line, its four 16-byte loads issued together (one round trip)
        gptr_d2 lp = (gptr_d2)(uintptr_t)(td.leaves + leaf[j]);
        const vd2 la = lp[0], lb = lp[1], lc = lp[2], ld = lp[3];
#ifdef MADICP_STAMPS
        if (u == u_first && base == r * S) { MADICP_STAMP_WAIT(); MADICP_STAMP(8); }
        if (u == u_first && base == r * S + QPT * kBlock) { MADICP_STAMP_WAIT(); MADICP_STAMP(12); }
#endif
        // gate (mad_icp.cpp:81-83)
        const double g0 = q0[j] - la.x, g1 = q1[j] - la.y, g2 = q2[j] - lb.x;
        const double src_ball = min_ball + b_ratio * pn[j];
        // (deciding from the squares and taking the root only within 2^-50 of the threshold was built and measured: +4.6 us per
        // launch at BASELINE configs[4], nothing at the headline — the wave-wide vote costs more than the root; profiles/r5_e_ab.md)
        const double dist = sqrt(dotc(g0, g1, g2, g0, g1, g2));
        const bool rejected = dist > src_ball;
        // the pair's threshold: its leaf's — and, rejected, no more than how far outside its ball it is (gate reuse), marked by
        // the sign; stored only when it changes (an accepted pair that keeps its leaf is never written)
        if (cache_margin) {
          float tnew = tkeep[j];
          if (rejected && gate_file) tnew = -fminf(tkeep[j], __double2float_rd((dist - src_ball) + wearv[j]));
          if (walk[j] || tnew != cmar[j]) cache_margin[(long long)k * L + i] = tnew;
        }
        if (TRACE && corr) corr[(long long)td.slot * L + i] = static_cast<uint32_t>(leaf[j]) | (rejected ? 0x80000000u : 0u);
        if (rejected) continue;
        if (mark_matched) matched[i] = 1;  // idempotent byte store (mad_icp.cpp:85)

        const double bbox0 = ld.x;
        const double n0 = lb.y, n1 = lc.x, n2 = lc.y;
#ifndef MADICP_EXACT_SOLVE
        // From here on nothing decides a branch of the reference (the gate above was the last decision; the Huber
        // switch below is continuous in e), and the kernel is bound by the INSTRUCTIONS its twelve waves issue: the
        // residual, the Jacobian and the 27 accumulations use fused multiply-adds (one instruction where the
        // reference's order needs two), the zero columns of skew(p) are not multiplied out, and the two quotients use
        // the refined reciprocal (1/min_ball once per kernel).  (H, b) differ from the reference-order sums in the last
        // bits — they already do by the order of the reduction; the pose contract is 1e-5.
        {
#pragma clang fp contract(fast)
          // errorAndJacobian (mad_icp.cpp:59-72)
          const double e = g0 * n0 + g1 * n1 + g2 * n2;
          double J[6];
          J[0] = n0 * R[0] + n1 * R[3] + n2 * R[6];
          J[1] = n0 * R[1] + n1 * R[4] + n2 * R[7];
          J[2] = n0 * R[2] + n1 * R[5] + n2 * R[8];
          // -J[0:3] * skew(p)
          J[3] = J[2] * py[j] - J[1] * pz[j];
          J[4] = J[0] * pz[j] - J[2] * px[j];
          J[5] = J[1] * px[j] - J[0] * py[j];
          // Huber x planarity weight (mad_icp.cpp:92-98; `abs` there is fabs — SURVEY fact 4)
          double scale = 1.0;
          const double chi = fabs(e);
          if (chi > rho) scale = fast_div(rho, chi, fast_rcp(chi));
          const double w = 1.0 - fast_div(bbox0, min_ball, inv_min_ball);
          scale *= w * w;
          double sJ[6];
#pragma unroll
          for (int rr = 0; rr < 6; ++rr) sJ[rr] = scale * J[rr];
          int v = 0;
#pragma unroll
          for (int cc = 0; cc < 6; ++cc)
#pragma unroll
            for (int rr = cc; rr < 6; ++rr) {
              acc[v] = __builtin_fma(sJ[rr], J[cc], acc[v]);
              ++v;
            }
#pragma unroll
          for (int rr = 0; rr < 6; ++rr) acc[21 + rr] = __builtin_fma(sJ[rr], e, acc[21 + rr]);
        }
#else

        // errorAndJacobian (mad_icp.cpp:59-72)
        const double e = dotc(g0, g1, g2, n0, n1, n2);
        double J[6];
        J[0] = dotc(n0, n1, n2, R[0], R[3], R[6]);
        J[1] = dotc(n0, n1, n2, R[1], R[4], R[7]);
        J[2] = dotc(n0, n1, n2, R[2], R[5], R[8]);
        // -J[0:3] * skew(p): columns of skew(p) are (0,pz,-py), (-pz,0,px), (py,-px,0)
        const double a0 = -J[0], a1 = -J[1], a2 = -J[2];
        J[3] = dotc(a0, a1, a2, 0.0, pz[j], -py[j]);
        J[4] = dotc(a0, a1, a2, -pz[j], 0.0, px[j]);
        J[5] = dotc(a0, a1, a2, py[j], -px[j], 0.0);

        // Huber x planarity weight (mad_icp.cpp:92-98; `abs` there is fabs — SURVEY fact 4)
        double scale = 1.0;
        const double chi = fabs(e);
        if (chi > rho) scale = rho / chi;
        const double w = 1.0 - bbox0 / min_ball;
        scale *= w * w;

        double sJ[6];
#pragma unroll
        for (int rr = 0; rr < 6; ++rr) sJ[rr] = scale * J[rr];
        int v = 0;
#pragma unroll
        for (int cc = 0; cc < 6; ++cc)
#pragma unroll
          for (int rr = cc; rr < 6; ++rr) acc[v++] += sJ[rr] * J[cc];
#pragma unroll
        for (int rr = 0; rr < 6; ++rr) acc[21 + rr] += sJ[rr] * e;
#endif
        acc[27] += 1.0;
      }
      if (u == u_first && base == r * S) { MADICP_STAMP(9); }
    }
  }
