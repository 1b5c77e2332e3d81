// Pass policy of the data pass: unit costs, the thresholds that follow from them, and their calibration.
//
// gmmvb_estep / gmmvb_mstep choose between kernels whose results agree to rounding; only the time depends on the choice.
// Every number the choice uses is one of the unit costs below or a threshold derived from them; the few that are plain
// observations cite the measurement.  The literals were measured on MI355X at the benchmark shape (K 64, D 128: bench.py's
// `dense` leg, profiles/r4_bench_line_w5s20.json, the kernel trace profiles/r4_bench_kernel_summary.md, the spread sweep,
// profiles/r*_experiments.md); a workspace scales them to its own shape (tile counts) when it is created, and - round 6 -
// replaces the three bulk costs by what ITS OWN first dense E-step, dense M-step and bound pass take on ITS device
// (gmmvb_policy_calibrate: HIP events around those launches, read back like the pass counters; the literals stay in force
// until a measurement has arrived and whenever one falls outside [1/2, 2] x the scaled literal).  The costs that are not
// measured follow their sibling on the same pipe (exact pairs and list M-step: the dense f64 kernels; proof pairs: the int8
// bound pass).  gmmvb_policy_table reads the table.
#pragma once
#include "common.h"

namespace gmmvb {

struct PolicyTable {
    // ---- per block / tile pair, in units of 1e-11 s (they enter the cost formulas multiplied by tile counts)
    double i8_block_pair = 0.12;     // int8 bound pass, per 32 x 32 block pair of a (row, component) pair
    double i8_row_of_y = 0.039;      // ... per row of y its epilogue bounds (32 per output block)
                                     //   all four blocks at D = 128: 0.12 * 10 + 0.039 * 128 = 6.2 -> 0.062 ns per pair = the
                                     //   38-ms pass over 6.4e8 pairs of the kernel trace
    double f64_tile_pair = 0.81;     // exact evaluation (estep_gather_dev_f64) per 16 x 16 f64 tile pair: 36 of them at D = 128
                                     //   = 0.29 ns per pair; trace: 0.31-0.33 ns per pair in bulk
    double proof_per_exact = 0.33;   // an int8 proof pair (0.10 ns, tools/bench_proof.py) in exact pairs (0.31 ns)
    // ---- ns per (row, component) pair at THIS workspace's shape (literals: D = 128, scaled by tile counts in init())
    double dense_e_ns = 0.269, dense_m_ns = 0.261;      // dense kernels: E 172.0 ms / 6.4e8, M 167.1 ms / 6.4e8 (`dense` leg)
    double bound_ns = 0.062 + 0.010;                    // bound pass + record building / selection around it (6 of 44 ms)
    double exact_ns = 0.31, proof_ns = 0.10;
    double list_m_ns = 0.355;        // list M-step per accumulated pair on long lists (22 of 64 active: 78 ms / 2.2e8,
                                     //   profiles/r3_full_run.json pass 2; 0.29-0.30 on short lists since round 4)
    // ---- observations (not unit costs)
    // An overflow row (no usable reference: all K pairs evaluated) costs K exact_ns against K bound_ns for bounding it afresh, so
    // carrying stops paying at 0.072 / 0.31 = 0.23 overflow rows per row - and overflow rows multiply by 4-8 from one carried
    // pass to the next (profiles/r2_experiments.md): 0.23 / 8 = 0.029, rounded down
    double overflow_rows = 0.02;
    // Spare candidates (listed, then found inactive) of a carried pass grow by about 2.5x per pass (same source): carrying goes on
    // while evaluating next pass's spares costs less than a fresh bound pass
    double spare_growth = 2.5;
    // A carried pass that evaluates more than this share of the pairs has lost its bounds (a fresh bound pass at the benchmark
    // shape leaves 0.05-0.10: profiles/r3_experiments.md, "bound level" rows)
    double carried_eval_above = 0.35;
    // The carry u' = c' - (gamma d - delta)^2 / 2 keeps gamma^2 of a pair's distance: below gamma = 0.5 a pair four thresholds
    // away becomes a candidate - nothing survives; straight from a dense pass (parameters still jumping) the measured limit is
    // higher: gamma < 0.85 left 118 of 256 candidates per row at config 4 (171 ms, profiles/r3_experiments.md)
    double gamma_no_carry = 0.5, gamma_no_carry_after_dense = 0.85;
    // A sweep straight after a dense pass carries K exact values per row: only worth it when few are active
    double sweep_after_dense_below = 0.1;
    // Regrouping the rows by dominant component (8 ms at the benchmark shape) pays once the passes are list-driven: at most 2.5
    // active components per row to force the one regrouping bound pass, at most 4 to regroup at a bound pass that happens
    // anyway, again after 5 % of the rows have changed their component (profiles/r2_experiments.md: gather + select 7.3 -> 6.4 ms,
    // list M-step 4.8 -> 4.2 ms on grouped rows; profiles/r3_experiments.md r3a2: regrouping at up to 32 active was worse)
    double regroup_force_below = 2.5, regroup_below = 4.0, regroup_moved = 0.05;
    // The own-pair round before the sweep is skipped while no component moves: own_first() needs Gamma > 1.004 or delta > 0.04,
    // which min_k (gamma_k - delta_k / 30) >= 0.995 rules out for every k (1 / 1.004 = 0.996; 0.04 / 30 = 0.0013)
    double own_round_below = 0.995;
    // ---- calibration state
    double lit_dense_e = 0.0, lit_dense_m = 0.0, lit_bound = 0.0, lit_exact = 0.0, lit_proof = 0.0, lit_list_m = 0.0;   // scaled literals
    int measured = 0;                // bit 0 dense E, 1 dense M, 2 bound pass: taken from this workspace's own passes

    // The literals at a shape of T f64 feature tiles / t32 int8 feature blocks: every per-pair cost is a sum over tile pairs
    void init(int T, int t32) {
        const double f = tri_pairs(T) / 36.0;
        lit_dense_e = dense_e_ns = 0.269 * f;
        lit_dense_m = dense_m_ns = 0.261 * f;
        lit_exact = exact_ns = f64_tile_pair * tri_pairs(T) / 100.0 * (0.31 / 0.2916);      // (0.31 measured in bulk against 0.29 modelled)
        lit_proof = proof_ns = proof_per_exact * exact_ns;
        lit_list_m = list_m_ns = 0.355 * f;
        lit_bound = bound_ns = (t32 > 0 ? (i8_block_pair * tri_pairs(t32) + i8_row_of_y * 32 * t32) / 100.0 : 0.062 * f) + 0.010 * f;
        measured = 0;
    }
    // one measurement (ns per pair) of what: 0 dense E, 1 dense M, 2 bound pass.  Returns whether it was taken.
    bool take(int what, double ns) {
        const double lit = what == 0 ? lit_dense_e : (what == 1 ? lit_dense_m : lit_bound);
        if (!(ns >= 0.5 * lit && ns <= 2.0 * lit)) return false;         // (also NaN) a disturbed pass: the literal stays
        if (what == 0) {
            dense_e_ns = ns;
            exact_ns = lit_exact * (ns / lit_dense_e);                   // the same f64 matrix pipe, the same clocks
        } else if (what == 1) {
            dense_m_ns = ns;
            list_m_ns = lit_list_m * (ns / lit_dense_m);
        } else {
            bound_ns = ns;
            proof_ns = lit_proof * (ns / lit_bound);                     // the same int8 pipe
        }
        measured |= 1 << what;
        return true;
    }
    // The pruned E-step pays the bound pass for every pair and proof + exact evaluation for the active ones:
    //   bound_ns K + act (exact_ns + proof_ns)  <  dense_e_ns K   <=>   act / K < (0.269 - 0.072) / 0.41 = 0.48 at the literals
    double prune_below() const { return (dense_e_ns - bound_ns) / (exact_ns + proof_ns); }
    // ... and once a bound pass has left `eval` pairs per row for the exact kernels, the dense kernel is the cheaper next pass if
    //   bound_ns K + eval exact_ns > dense_e_ns K   <=>   eval / K > 0.64 (the proof round has run by then: its cost is sunk)
    double dense_again_above() const { return (dense_e_ns - bound_ns) / exact_ns; }
    // The list M-step wins while act list_m_ns < K dense_m_ns  <=>  act / K < 0.73; a sixth off for building lists that long: 0.6.
    // (Round 3 used 0.35; the spread sweep's spread 0.75 - 31 to 40 of 64 components active for twenty passes - is where it matters.)
    double list_m_below() const { return dense_m_ns / list_m_ns * (5.0 / 6.0); }
};

}  // namespace gmmvb
