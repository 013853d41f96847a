// LEAF-MAJOR ROUNDS — the round of a DEEP launch (a batch shares the chip: few workgroups per scan, every workgroup owns
// several keyframe trees) in which the pairs that still have to walk are few.  TEXTUALLY included by icp_round in front of the
// tree-major body (icp_linearize_body.inc.h), live only in the DEEP = true instantiation; same names in scope.
//
// What the per-round trace of BASELINE configs[4] (64 keyframes, 8 scans in flight) said about the tree-major order
// (profiles/r5_k64_rounds.md): a round in which NOTHING walks costs 90-100 us and neither skipping the rejected pairs' leaf
// records (gate reuse) nor dropping the per-round margin stores moved it — it reads 44 bytes per (leaf, tree) pair, 529 MB per
// launch = 5.9 TB/s: the moving leaf (x, y, z, |p|: 32 B) once per TREE, and 12 B of cached correspondence.  And a round in
// which 0.1-10 % of the pairs walk costs 150-300 us, because a wavefront that holds ONE walker waits out a whole descent.
//
// Here the launch geometry gives every workgroup ONE range of the scan's leaves and ALL the trees of its XCD piece
// (ranges_per_tree = workgroups per piece, pick_geometry), and the loops are turned inside out:
//     for every pass of 768 leaves:  p, |p|, q = X p ONCE;  for every tree of the piece:  cached correspondence (12 B, the
//     loads of four trees in flight together) -> reuse test -> gate reuse or leaf record, gate, e, J, accumulation
// so the moving leaf is read and transformed once per 8 trees (32 + 18 flops -> 4 B + 2 flops per pair at 64 keyframes), and
// a pair that has to walk is not walked in place but QUEUED — per wavefront, 2 bytes in LDS — and the queue is walked DENSELY
// (64 walkers per descent, from global memory: the LDS-staged top belongs to one tree) when it is full or the range is done;
// the walker's leaf record, gate and accumulation follow its descent.  The number of descents a wavefront waits for is
// ceil(walkers / 64), not the number of passes that hold one.  Round 6: the pairs that keep their leaf and have to be EVALUATED
// are queued the same way (pass | tree | lane, leaf ordinal, threshold on file: 10 bytes in LDS) and evaluated 64 at a time as
// soon as 64 are there — where they stood, a fifth of a wavefront's lanes per tree were active through the two hundred fp64
// instructions of gate, Jacobian and 28 accumulations (the counters of profiles/r6_k64_pmc_summary.md: the vector ALU is the
// busiest unit of this kernel), now all 64 are.
//
// The ACCUMULATION ORDER is another one than the tree-major body's (queue order: pass, tree, lane, over the lanes of a wavefront as
// they come; walkers last), so H and b differ from it in
// their last bits (~1e-16 relative; the pose contract is 1e-5 and the reference's own order depends on its thread count) —
// every DECISION (leaf, depth, gate, matched flag, visit count) is the same on the data the tests hold it to — structurally so
// except at exact ties: the pose of a later round differs from the tree-major one's by ~1e-16, so a pair that sits EXACTLY on a
// split plane or on the gate's radius can fall on the other side — and the launch is deterministic: which
// rounds run leaf-major is decided from the hint the previous round left (nodes walked), itself a function of the inputs.
// tests/test_gpu_parity.py::test_deep_launches_leaf_major_rounds holds exactly that.  Chosen per round, per workgroup, without a
// vote: not round 0, the workgroup walked fewer than `leaf_major` nodes per pass last round (the option; default 8192 of the
// ~10 400 a pass of 768 walkers visits: measured at 64 keyframes x 8 scans, avg launch 180 us never, 159 us at 512, 149 at 2 048,
// 133 at 8 192 — the dense queue beats the in-pass walk even when a third of the pairs walk), every unit of the workgroup is the
// same range (geometry), at most kDeepTrees trees and kDeepPasses passes.
  if (DEEP && QPT == 1 && leaf_major) {
    const int q_lane = MADICP_TID & 63, q_wave = MADICP_TID >> 6;
    const int n_my = (hi - u_first + nslots - 1) / nslots;  // trees of this workgroup: k_first .. k_first + n_my - 1
    // their descriptors -> LDS (11 eight-byte words each)
    for (int w = MADICP_TID; w < n_my * 11; w += kBlock) {
      const int tt = w / 11, ww = w - 11 * tt;
      reinterpret_cast<long long*>(&s_tds[tt])[ww] =
          ((const __attribute__((address_space(1))) long long*)(uintptr_t)&job->trees[k_first + tt])[ww];
    }
    __syncthreads();
    const int i_lo = r_first * S, i_hi = min(Lv, (r_first + 1) * S);  // (virtual indices: kernels.hip.h, "Ranges")
    int qn = 0;  // entries in this wavefront's queue (wave-uniform)

    // the evaluation of one pair whose leaf is known: record, gate (mad_icp.cpp:81-83), e, J, weights, accumulation
    // (mad_icp.cpp:59-101) — the arithmetic of the tree-major body, statement for statement
    // (t_file: the pair's threshold on file, t_keep: its leaf's own — kernels.hip.h, "One threshold per pair")
    // (la .. ld: the leaf's record, one 64-byte line as four 16-byte loads, requested by the caller)
    auto evaluate = [&](const vd2 la, const vd2 lb, const vd2 lc, const vd2 ld, int kk, int i, bool walked_now, float t_file, float t_keep,
                        double wear, double px, double py, double pz, double pnorm, double q0, double q1, double q2) {
      const double g0 = q0 - la.x, g1 = q1 - la.y, g2 = q2 - lb.x;
      const double src_ball = min_ball + b_ratio * pnorm;
      const double dist = sqrt(dotc(g0, g1, g2, g0, g1, g2));
      const bool rejected = dist > src_ball;
      {
        float tnew = t_keep;
        if (rejected && gate_file) tnew = -fminf(t_keep, __double2float_rd((dist - src_ball) + wear));
        if (walked_now || tnew != t_file) cache_margin[(long long)kk * L + i] = tnew;
      }
      if (rejected) return;
      if (mark_matched) matched[i] = 1;
      const double bbox0 = ld.x;
      const double n0 = lb.y, n1 = lc.x, n2 = lc.y;
#ifndef MADICP_EXACT_SOLVE
      {
#pragma clang fp contract(fast)
        const double e = g0 * n0 + g1 * n1 + g2 * n2;
        double J[6];
        J[0] = n0 * R[0] + n1 * R[3] + n2 * R[6];
        J[1] = n0 * R[1] + n1 * R[4] + n2 * R[7];
        J[2] = n0 * R[2] + n1 * R[5] + n2 * R[8];
        J[3] = J[2] * py - J[1] * pz;
        J[4] = J[0] * pz - J[2] * px;
        J[5] = J[1] * px - J[0] * py;
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
      const double e = dotc(g0, g1, g2, n0, n1, n2);
      double J[6];
      J[0] = dotc(n0, n1, n2, R[0], R[3], R[6]);
      J[1] = dotc(n0, n1, n2, R[1], R[4], R[7]);
      J[2] = dotc(n0, n1, n2, R[2], R[5], R[8]);
      const double a0 = -J[0], a1 = -J[1], a2 = -J[2];
      J[3] = dotc(a0, a1, a2, 0.0, pz, -py);
      J[4] = dotc(a0, a1, a2, -pz, 0.0, px);
      J[5] = dotc(a0, a1, a2, py, -px, 0.0);
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
    };

    // the queue of this wavefront, walked densely: every lane one queued pair (pass, tree, lane of origin)
    auto drain = [&]() {
      wave_lds_order();
      for (int b = 0; b < qn; b += 64) {
        const bool has = b + q_lane < qn;
        const int e = has ? (int)s_queue[q_wave][b + q_lane] : 0;
        const int tt = (e >> 6) & (kDeepTrees - 1), pc = e >> (6 + kDeepTreesLog2);
        const int i = phys(r_first, i_lo + pc * kBlock + q_wave * 64 + (e & 63));
        const TreeDesc& td = s_tds[has ? tt : 0];
        const vd4 p = has ? ((gptr_d4)(uintptr_t)moving)[i] : vd4{0.0, 0.0, 0.0, 0.0};
#ifdef MADICP_XFORM_HOMOGENEOUS
        const double a0[1] = {((R[0] * p.x + R[1] * p.y) + R[2] * p.z) + t[0]};
        const double a1[1] = {((R[3] * p.x + R[4] * p.y) + R[5] * p.z) + t[1]};
        const double a2[1] = {((R[6] * p.x + R[7] * p.y) + R[8] * p.z) + t[2]};
#else
        const double a0[1] = {t[0] + dots(R[0], R[1], R[2], p.x, p.y, p.z)};
        const double a1[1] = {t[1] + dots(R[3], R[4], R[5], p.x, p.y, p.z)};
        const double a2[1] = {t[2] + dots(R[6], R[7], R[8], p.x, p.y, p.z)};
#endif
        const bool wv[1] = {has};
        int xi[1], xl[1], xd[1];
        double xm[1] = {3.0e38};
        descend_multi<1>(td, s_top, s_exit, 0, a0, a1, a2, wv, xi, xl, xd, xm);  // (global memory: no staged top here)
        if (has) {
          const int kk = k_first + tt;
          const double wear = __builtin_fma(p.w, wear_alpha, wear_beta + (double)round * (1e-11 * (td.rho + fabs(td.origin[0]) +
                                                                                                    fabs(td.origin[1]) + fabs(td.origin[2]) + 1.0)));
          visits += (unsigned int)xd[0];
          walked_visits += (unsigned int)xd[0];
          const long long ci = (long long)kk * L + i;
          const bool cacheable = xd[0] <= kCacheMaxDepth && (unsigned int)xl[0] <= kCacheIdxMask;
          cache_leaf[ci] = (unsigned int)xl[0] | ((unsigned int)xd[0] << 26);
          gptr_d2 lp = (gptr_d2)(uintptr_t)(td.leaves + xl[0]);
          const vd2 la = lp[0], lb = lp[1], lc = lp[2], ld = lp[3];
          evaluate(la, lb, lc, ld, kk, i, true, 0.f, cacheable ? __double2float_rd(xm[0] + wear) : 0.f, wear, p.x, p.y, p.z, p.w, a0[0],
                   a1[0], a2[0]);
        }
      }
      qn = 0;
      wave_lds_order();
    };

    // the pairs that keep their leaf and have to be evaluated: not where they stand — at BASELINE configs[4] a fifth of a wavefront's
    // lanes per tree, and the evaluation is two hundred fp64 instructions for all 64 — but queued like the walkers (in pass, tree,
    // lane order: deterministic) and evaluated 64 at a time: the lane that evaluates an entry fetches the leaf's coordinates and
    // its record together (the entry carries the leaf ordinal), transforms, evaluates.  count <= 64 entries from the head; the
    // rest moves to the front.
    int qe = 0;
    auto eval_flush = [&](int count) {
      wave_lds_order();
      const bool has = q_lane < count;
      const int rest = qe - count;  // (< 64)
      const int e = has ? (int)s_eq_id[q_wave][q_lane] : 0;
      const unsigned int lf = has ? s_eq_leaf[q_wave][q_lane] : 0u;
      const float t_file = has ? s_eq_t[q_wave][q_lane] : 0.f;
      const bool mv = q_lane < rest;
      const unsigned short m_id = mv ? s_eq_id[q_wave][count + q_lane] : (unsigned short)0;
      const unsigned int m_lf = mv ? s_eq_leaf[q_wave][count + q_lane] : 0u;
      const float m_t = mv ? s_eq_t[q_wave][count + q_lane] : 0.f;
      wave_lds_order();
      if (mv) {
        s_eq_id[q_wave][q_lane] = m_id;
        s_eq_leaf[q_wave][q_lane] = m_lf;
        s_eq_t[q_wave][q_lane] = m_t;
      }
      qe = rest;
      const int tt = (e >> 6) & (kDeepTrees - 1), pce = e >> (6 + kDeepTreesLog2);
      const int i = phys(r_first, i_lo + pce * kBlock + q_wave * 64 + (e & 63));
      const TreeDesc& td = s_tds[has ? tt : 0];
      vd4 p = vd4{0.0, 0.0, 0.0, 0.0};
      vd2 la = vd2{0.0, 0.0}, lb = la, lc = la, ld = la;
      if (has) {  // (coordinates and record are independent loads: one round trip)
        p = ((gptr_d4)(uintptr_t)moving)[i];
        gptr_d2 lp = (gptr_d2)(uintptr_t)(td.leaves + (int)lf);
        la = lp[0]; lb = lp[1]; lc = lp[2]; ld = lp[3];
      }
#ifdef MADICP_XFORM_HOMOGENEOUS
      const double e0 = ((R[0] * p.x + R[1] * p.y) + R[2] * p.z) + t[0];
      const double e1 = ((R[3] * p.x + R[4] * p.y) + R[5] * p.z) + t[1];
      const double e2 = ((R[6] * p.x + R[7] * p.y) + R[8] * p.z) + t[2];
#else
      const double e0 = t[0] + dots(R[0], R[1], R[2], p.x, p.y, p.z);
      const double e1 = t[1] + dots(R[3], R[4], R[5], p.x, p.y, p.z);
      const double e2 = t[2] + dots(R[6], R[7], R[8], p.x, p.y, p.z);
#endif
      if (has) {
        const double wear = __builtin_fma(p.w, wear_alpha, wear_beta + (double)round * (1e-11 * (td.rho + fabs(td.origin[0]) +
                                                                                                  fabs(td.origin[1]) + fabs(td.origin[2]) + 1.0)));
        evaluate(la, lb, lc, ld, k_first + tt, i, false, t_file, fabsf(t_file), wear, p.x, p.y, p.z, p.w, e0, e1, e2);
      }
      wave_lds_order();
    };

    int pc = 0;
    for (int base = i_lo; base < i_hi; base += kBlock, ++pc) {
      // (a pass queues at most 64 walkers per tree and wavefront: room is made HERE, where no pair is in registers)
      if (qn + 64 * n_my > kQueueCap) drain();
      const int i = phys(r_first, base + MADICP_TID);
      const bool valid = base + MADICP_TID < i_hi && i < L;
      vd4 p = vd4{0.0, 0.0, 0.0, 0.0};
      if (pc == 0) p = pv0[0];  // (fetched before the solve prologue; clamped index: harmless for an invalid lane)
      else if (valid) p = ((gptr_d4)(uintptr_t)moving)[i];
#ifdef MADICP_XFORM_HOMOGENEOUS
      const double q0 = ((R[0] * p.x + R[1] * p.y) + R[2] * p.z) + t[0];
      const double q1 = ((R[3] * p.x + R[4] * p.y) + R[5] * p.z) + t[1];
      const double q2 = ((R[6] * p.x + R[7] * p.y) + R[8] * p.z) + t[2];
#else
      const double q0 = t[0] + dots(R[0], R[1], R[2], p.x, p.y, p.z);
      const double q1 = t[1] + dots(R[3], R[4], R[5], p.x, p.y, p.z);
      const double q2 = t[2] + dots(R[6], R[7], R[8], p.x, p.y, p.z);
#endif
      for (int t0 = 0; t0 < n_my; t0 += 4) {  // the trees of this workgroup, four at a time: their cached records in flight together
        unsigned int cw[4];
        float cm[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          cw[a] = 0u; cm[a] = 0.f;
          if (valid && t0 + a < n_my) {
            const long long ci = (long long)(k_first + t0 + a) * L + i;
            cw[a] = ((const __attribute__((address_space(1))) unsigned int*)(uintptr_t)cache_leaf)[ci];
            cm[a] = ((const __attribute__((address_space(1))) float*)(uintptr_t)cache_margin)[ci];
          }
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          if (t0 + a >= n_my) break;  // (uniform)
          const int tt = t0 + a;
          const TreeDesc& td = s_tds[tt];
          // (ONE expression for a pair's wear in all three places — the tree-major body, the drain above, here — so that the stored
          // thresholds and the walked counter are the same doubles whichever mode a round ran in)
          const double wear = __builtin_fma(p.w, wear_alpha, wear_beta + (double)round * (1e-11 * (td.rho + fabs(td.origin[0]) +
                                                                                              fabs(td.origin[1]) + fabs(td.origin[2]) + 1.0)));
          const bool keep = valid && (double)fabsf(cm[a]) > wear;
          const bool w = valid && !keep;
          // queue the walkers: pass, tree, lane — in pass, tree, lane order (a ballot and a prefix count: deterministic)
          const unsigned long long wm = __ballot(w);
          if (wm) {  // (wave-uniform)
            if (w) s_queue[q_wave][qn + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(wm >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)wm, 0u))] =
                       (unsigned short)((pc << (6 + kDeepTreesLog2)) | (tt << 6) | q_lane);
            qn += __popcll(wm);
            walked |= w;
          }
          if (keep) visits += cw[a] >> 26;
          // (a negative threshold above the wear: same leaf, still rejected — nothing to evaluate)
          const bool ev = keep && !(gate_reuse && cm[a] < 0.f);
          const unsigned long long em = __ballot(ev);
          if (em) {  // (wave-uniform)
            if (ev) {
              const int at = qe + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(em >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)em, 0u));
              s_eq_id[q_wave][at] = (unsigned short)((pc << (6 + kDeepTreesLog2)) | (tt << 6) | q_lane);
              s_eq_leaf[q_wave][at] = cw[a] & kCacheIdxMask;
              s_eq_t[q_wave][at] = cm[a];
            }
            qe += __popcll(em);
            if (qe >= 64) eval_flush(64);
          }
        }
      }
    }
    if (qe > 0) eval_flush(qe);
    drain();
  } else
