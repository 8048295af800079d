| what | number |
|---|---|
| **headline**: `NeuralSemiCRFInterval(score, noise).logProb(intervals)` forward + backward through the public API, T=1024 × 352 | **__HEADLINE__ steps/s** (__MS__ ms per step; five repetitions __REPS__) |
| log-partition forward sweep, T=1024 × 352 (the roofline kernel) | __FWDUS__ µs = __ACH__ GB/s algorithmic = **__FRAC__ of the 8 TB/s HBM roofline**; HBM traffic 1.07 × the algorithmic bytes |
| gradient sweep, T=1024 × 352 | 340 µs minimum, 353 average over 23 launches (the boxes of the round: 320–340 minimum) (0.74 GB read + 0.74 GB written; the dense gradient's zeros are written once per pooled buffer) |
| BASELINE config #2, T=1024 × 88: forward sweep / gradient sweep / `logProb` forward + backward | __C2FWD__ µs (__C2FRAC__) / 185 µs / __C2MS__ ms |
| decode T=2048 × 352, `forcedStartPos` set: "decode segments/s" on the device / as packed arrays / as the reference's Python lists | __DECDEV__ (__DECMS__ ms) / __DECPK__ ms / __DECAPI__ segments/s (__DECLIST__ ms: 657 k tuples at the CPython floor) |
| interval scorer T=1024 × 352 × D=256, exact fp32: forward / backward | __SF__ ms (__SFR__ of the fp32 matrix pipe) / __SB__ ms |
| ... opt-in three-limb bf16 (`scorer.contraction`, fp32-grade, not bit-identical): forward / backward | __SF3__ ms / __SB3__ ms (__S3F__ / __S3B__ of the 2.5 PFLOP/s bf16 pipe) |
| one real segment (T=691, 90 symbols): scorer + CRF `logProb` forward + backward, fused route / as the reference calls it | __SEG1__ / __SEG1U__ ms |
| four segments, fused route | __SEG4__ ms |
| train-shaped data-parallel step (4 × 90 × 691 per rank): exact / `"bf16x3-train"` / `"bf16x3-all"` | __TR__ / __TRT__ / __TRA__ ms |
| transcription loop, one / four recordings | __TL1__ / __TL4__ segments/s end to end |
| CPU baseline on the same box (the reference's op loop as a torch-CPU port, __CPUC__ threads; this library's host kernels, 32 threads) | __CPU__ / __CPUH__ steps/s |

