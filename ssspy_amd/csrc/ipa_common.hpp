// Scalar pieces of the IPA / LQPQM kernels shared by the lane-per-bin kernels (ipa_kernels.hip) and
// the 8-lanes-per-bin ones (ipa_rows.hip).
#pragma once

#include "common.hpp"

namespace ssspy {

// mode of a source step: NEWTON_FIXED max_iter steps; NEWTON_PROBE max_iter steps, convergence bits
// AND-ed into the mixture's word (nothing else is produced); NEWTON_APPLY the number of steps found
// in the word by k_newton_steps
enum { NEWTON_FIXED = 0, NEWTON_PROBE = 1, NEWTON_APPLY = 2, NEWTON_FUSED = 3 };

__device__ __forceinline__ double floor_of_zero(int floor_kind, double eps) {
  return apply_floor(0.0, floor_kind, eps);
}

// largest real root of x^3 + A x^2 + B x + C, computed the way the reference does (complex
// Cardano with the principal polar cube root, whose real part is kept even when it is not the real
// root: the value only seeds the Newton iteration and is range-checked afterwards)
__device__ __forceinline__ double largest_cubic_root(double A, double B, double C) {
  const double P = -(A * A) / 3.0 + B;
  const double Q = (2.0 * A * A * A) / 27.0 - (A * B) / 3.0 + C;
  const double disc = (Q * 0.5) * (Q * 0.5) + (P / 3.0) * (P / 3.0) * (P / 3.0);
  // w = -Q/2 + sqrt(disc) (principal complex square root)
  const double wr = -0.5 * Q + (disc >= 0.0 ? sqrt(disc) : 0.0);
  const double wi = disc >= 0.0 ? 0.0 : sqrt(-disc);
  const double mag = sqrt(wr * wr + wi * wi);
  double ur, ui, vr, vi, x1;
  if (mag == 0.0) {
    ur = 1.0;
    ui = 0.0;
    vr = -P / 3.0;
    vi = 0.0;
    x1 = cbrt(-Q);
  } else {
    const double m3 = cbrt(mag), th = atan2(wi, wr) / 3.0;
    ur = m3 * cos(th);
    ui = m3 * sin(th);
    // V = -P / (3 U)
    const double den = 3.0 * (ur * ur + ui * ui);
    vr = -P * ur / den;
    vi = P * ui / den;
    x1 = ur + vr;
  }
  double root = x1;
  if (P < 0.0 && !(disc > 0.0)) {
    const double h = 0.8660254037844386;  // sqrt(3)/2
    // Re(U w + V conj(w)), Re(U conj(w) + V w), w = (-1 + i sqrt(3)) / 2
    const double x2 = -0.5 * ur - h * ui - 0.5 * vr + h * vi;
    const double x3 = -0.5 * ur + h * ui - 0.5 * vr - h * vi;
    root = fmax(root, fmax(x2, x3));
  }
  return root - A / 3.0;
}

// Ordering inside a mixture's Newton vote.  Only two words cross workgroups there -- the AND-ed
// convergence bits and the arrival counter -- and both are touched with agent-scope atomics alone,
// which are performed at the memory side (they pass the XCD's L2): all the vote needs is that a
// workgroup's AND has been performed before its arrival is counted, and that the word is read
// after the count was seen -- a wait for the outstanding memory operations of the lane.  The
// agent-scope release / acquire fences that stood here wrote back and invalidated the XCD's whole
// L2 at every vote of every workgroup (cf. DESIGN.md section 4 item 53 iv).
__device__ __forceinline__ void vote_order() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

}  // namespace ssspy
