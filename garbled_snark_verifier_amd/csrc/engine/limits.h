// LDS budget shared by the program compiler (which assigns window slots) and the kernels (which lay LDS out).
// A CU has 160 KiB = 163,840 bytes of LDS; one workgroup takes all of it:
//   [0, 64 KiB)                   AES tables Te0 / Te2, bank-replicated, 256-byte entry stride (kernels.hip)
//   [64 KiB, +GSV_LDS_SLOTS*16)   label window
//   [.., +GSV_LDS_SLOTS)          plaintext bits of window wires (evaluate)
//   [.., +176)                    round keys
//   [.., +16)                     per-group step-barrier counters
#pragma once
#define GSV_LDS_TABLE_BYTES 65536u
#define GSV_LDS_SLOTS 5760u
#define GSV_LDS_RK_BASE (GSV_LDS_TABLE_BYTES + GSV_LDS_SLOTS * 16u + GSV_LDS_SLOTS)
#define GSV_LDS_GROUP_BAR_BASE (GSV_LDS_RK_BASE + 176u)  // four arrival counters of the per-group step barrier (kernels.hip)
#define GSV_LDS_BYTES (GSV_LDS_GROUP_BAR_BASE + 16u)
#if GSV_LDS_BYTES > 163840u
#error "LDS budget exceeded"
#endif
