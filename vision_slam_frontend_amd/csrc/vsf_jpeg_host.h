// vsf_jpeg_host.h -- what the host half (vsf_jpeg_host.cc: markers, tables, the upload plan; plain C++, parses UNTRUSTED
// bytes, built with AddressSanitizer by `make asan`) and the device half (k_jpeg.hip) of the JPEG decoder share: the
// layouts of one upload.
#ifndef VSF_JPEG_HOST_H_
#define VSF_JPEG_HOST_H_

#include <stdint.h>

namespace vsf_jpeg {

constexpr int kLookBits = 9;
constexpr int kMaxSlots = 6;     // distinct Huffman tables one image may use (3 components x DC / AC)
constexpr int kGroupBlocks = 16; // luminance blocks parked in LDS between two IDCT phases
constexpr int kSubBits = 16 - kLookBits;  // bits of a code beyond the first lookup
constexpr int kMaxSub = 12;      // second-level tables per Huffman table (the Annex K tables need 5 or 6)
constexpr uint32_t kLongCode = 0x8000u;
constexpr uint16_t kNoCode = 17 << 8;  // a prefix no code starts with: libjpeg's MAXCODE walk ends at its 17-bit sentinel and
                                       // answers symbol 0 -- seventeen bits are gone (jdhuff.c jpeg_huff_decode; damaged streams only)

struct DevHuff {                 // one Huffman table as the kernels read it
  uint16_t look[1 << kLookBits]; // 9-bit prefix -> (code length << 8 | symbol), or kLongCode | second-level table
  int32_t maxcode[18];           // T.81 F.2.2.3 (maxcode[17] = INT_MAX)
  int32_t valoff[17];            // VALPTR - MINCODE
  uint8_t vals[256];
  uint32_t nsub;                 // second-level tables in use; > kMaxSub: they do not fit (one-wave decoder only)
  uint16_t sub[kMaxSub][1 << kSubBits];  // the next 7 bits -> (code length << 8 | symbol)
};
static_assert(sizeof(DevHuff) == 1024 + 72 + 68 + 256 + 4 + kMaxSub * 256, "DevHuff layout");

struct DevTables {               // one distinct table set
  DevHuff huff[kMaxSlots];
  uint16_t qt_luma[64];          // natural order
};

struct DevImage {
  uint32_t stream_off;           // entropy-coded segment inside the packed stream buffer (4-byte aligned)
  uint32_t stream_len;
  uint32_t tables;               // index into the table sets
  int32_t ncomp, restart_interval, mcus_x, mcus_y;
  int32_t h[3], v[3];            // blocks per MCU of each component (1 x 1 for a single-component scan)
  int32_t dc_slot[3], ac_slot[3];
  int32_t par_ok;                // every Huffman table in use fits its second-level tables (the parallel decoder's need)
  int32_t n_scans;               // progressive files: the scans that carry the luminance component (0: a sequential file)
  uint32_t first_scan;           // ... and where they start in the upload's scan list
};

// Progressive files bring Huffman tables of their own with every scan (an encoder must optimise them: T.81 has no default
// progressive tables), ~5 per file: they travel as written in the DHT segment and a kernel expands them on the device.
struct DevHuffSrc {
  uint8_t bits[20];              // BITS[1..16] at [1..16]
  uint8_t vals[256];
};
struct DevHuffLite {             // what the one-wave decoders read of a table (no second-level tables)
  uint16_t look[1 << kLookBits];
  int32_t maxcode[18];
  int32_t valoff[17];
  uint8_t vals[256];
};
static_assert(sizeof(DevHuffSrc) == 276 && sizeof(DevHuffLite) == 1024 + 72 + 68 + 256, "progressive table layouts");

// One scan of a progressive file (T.81 Annex G) -- or of a sequential file whose components come in several scans: then
// Ss = 0, Se = 63 and a block is decoded whole (F.2.2) -- as the device walks it.  Only scans with the luminance component
// are listed: scans of chroma alone never touch what a gray read returns.
struct DevScan {
  uint32_t off, len;             // its entropy-coded segment, relative to the image's stream_off
  int32_t restart_interval;      // as the last DRI in front of the scan set it
  uint8_t ncomp, Ss, Se, Ah, Al; // components in the scan; spectral selection; successive approximation
  uint8_t comp[3];               // frame component (0 = luminance) of each scan component
  uint32_t huff[3];              // each scan component's Huffman table (DC scans: its DC table, AC scans: its AC table) in the
                                 // upload's list of progressive tables; unused in DC refinement scans
  uint32_t huff_ac[3];           // sequential scans (Ss = 0, Se = 63): the AC tables, huff[] holding the DC tables
  uint32_t pad_;
};
static_assert(sizeof(DevScan) == 48, "DevScan layout");

constexpr int kParThreads = 256;  // threads (= segments) of the parallel decoder per image
constexpr int kOverlap = 8;       // rows every segment's column carries past its end: the first rows of the next segment
constexpr int kTransSlack = kParThreads * (kOverlap + 2) * 4 + 2048;  // bytes the segment-major copy may exceed the stream by

}  // namespace vsf_jpeg
#endif  // VSF_JPEG_HOST_H_
