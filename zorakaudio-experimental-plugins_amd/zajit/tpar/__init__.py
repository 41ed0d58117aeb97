"""Time-parallel lowering of a leaf's @sample section: ONE WAVEFRONT PER INSTANCE, lane = frame.

The generic kernels (csrc/zab_generic.hip.h) run a script the way jsfx_process_block does (dsp_jsfx_aot.py:5785-5899): host
block after host block -- @block, the pending-mask @slider check -- and inside a block one frame after the other, one lane
per instance. Most of a dynamics / filter script is not serial in time at all, though. This module proves which parts are,
per leaf, and emits a second kernel `zab_<leaf>_tpar` that keeps the block structure (the wavefront runs @block and @slider
itself between the blocks, with the leaf's ordinary section code) and processes the frames of a block 64 at a time:

  1. @sample (user functions inlined, conditionals if-converted) becomes a DAG over the values of ONE frame: inputs
     spl0.., invariants (variables / sliders that @sample never writes: they change only in @block / @slider / @init, i.e.
     between blocks), constants, and `state-in` nodes -- the value a variable written by @sample had at the end of the
     PREVIOUS frame. A variable whose incoming value no path of @sample can observe (every read follows a write of the same
     frame) has no state-in at all: where a frame leaves it alone it carries a HOLD marker, and only its last written value
     is tracked.
  2. The cross-frame edges out(v)[t-1] -> state-in(v)[t] close cycles. Strongly connected components of that graph are
     the true recurrences; everything else is feed-forward in time and runs one lane per frame.
       * no cycle through state-in(v) ......... v is a delayed signal: a one-lane shift of out(v) (DPP wave_shr),
       * a cycle that is AFFINE in its states .. y[t] = A[t] y[t-1] + b[t] with A, b free of y (one-poles, leaky
         integrators, counters, sample-and-hold `c ? y = x`, biquads as 2x2): a weighted prefix scan over the wavefront
         with DPP row_shr / row_bcast moves (the scheme of the hand-written DDT kernel, csrc/kernels/ddt_fast.hip.h:98-108),
       * affine once its state-dependent conditions are fixed (attack/release smoothers, holds): fixed-point iteration of
         the condition pattern,
       * anything else: the minimal cycle runs as a uniform 64-step loop, inputs broadcast with v_readlane.
  3. loop() / while whose trip count is the same in every frame run as UNIFORM loops: trip k of all 64 frames together, then
     trip k + 1. Counters are wave-uniform, variables handed from trip to trip are per-lane values, and mem[] at addresses
     that depend on the counters only are PER-TRIP CELLS -- band k's filter state -- whose recurrence over the frames is
     classified and solved inside the trip exactly like a top-level one (the band loops of ERBTilt, SpectralStabilizer,
     EasyExpander, CMD). Reads at moving addresses inside such a loop are gathers (FIR taps into a ring: TSEQ, DOT).
  4. Values that depend on invariants only are computed once per host block.

The state a launch leaves in vars[] / spl[] / mem[] is what the serial path leaves. Affine components differ from the serial
order of operations by re-association only (O(1e-16) relative), checked by the same reference-VM fixtures as the generic
path (tests/test_tpar.py, tests/test_catalog_gpu.py). Everything the lowering assumes but cannot prove (distinct address
expressions address distinct cells, delay lines advance by one cell per frame, rare-event guards stay false) is checked at
run time; a launch that breaks an assumption is handed, from that point on, to the serial section code
(`zab_<leaf>_tpar_tail`).

`Plan.simulate` is a numpy restatement of the staged algorithm (one array element per lane) used by the CPU tests to pin
the analysis itself -- classification, coefficients, carries, partial chunks, loops -- without a GPU.

The package: `numeric` (value model), `graph` (the frame as a DAG, events), `plan` (recurrences, delay lines, schedule), `emit` (HIP
text), `simulate` (numpy restatement). Everything public is re-exported here.
"""
from .numeric import *      # noqa: F401,F403
from .graph import *        # noqa: F401,F403
from .plan import *         # noqa: F401,F403
from .emit import *         # noqa: F401,F403
from .simulate import *     # noqa: F401,F403
