// finish.hpp -- host side of the table build: turns the structured device results into the
// reference's byte streams (F1 <Exact>, F2 <Inexact>), applies libstdc++ std::sort where the
// reference does, and redoes lists with tied targets in the reference's discovery order.
// HTS = /root/reference/homTable_setup.cpp.
#pragma once
#include <stdint.h>
#include <functional>
#include <vector>

namespace cora {

struct HostGenome {
    int K = 0, d = 0, chr_short = 0;
    int n_groups = 0, n_threads = 0;
    uint8_t sig2[16];
    uint8_t sig4[256];
    std::vector<uint64_t> gstart;      // [n_contigs+1], last = G
    std::vector<uint32_t> len;         // [n_contigs]
    std::vector<uint32_t> window;      // [n_contigs] 0 or 50,000,000 (d == 3 sub-contig compaction)
    char tmpl[72];                     // seed template, 0 = unassigned (HTS:2015-2066)
    int seed_q = 8;
    // contig index (0-based) holding global position g
    uint32_t contig_of(uint64_t g) const;
};

// Seed template and per-seed signal offsets (HTS:2015-2116), shared by the host set-up.
void make_seed_template(int K, int d, char *tmpl /*[K]*/, int *seed_q);
void seed_signal_offsets(int K, const char *tmpl, int seed, int off[6] /* so, eo, ts1, te1, ts2, te2 */);
// Which contigs the reference compacts in 50 Mbp windows (HTS:727-765, 1467-1479, 2845-2860).
void plan_windows(int d, const std::vector<uint32_t> &len, std::vector<uint32_t> &window);

template <typename T>
struct View {                          // borrowed array (pinned host memory owned by the job)
    const T *p = nullptr;
    size_t n = 0;
    const T &operator[](size_t i) const { return p[i]; }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
};


// a list of <Inexact> that the host finishes (tied targets in its run, or more than 16 items)
struct HostNode {
    uint32_t node, count, tie, reserved;
    uint64_t stream_off;               // where its bytes belong in the device-written stream
    uint64_t item_off;                 // its items in the compacted host arrays
};

struct InexactDeviceResult {           // directed items sorted by (source, target)
    View<uint64_t> key;                // src<<32 | tgt
    View<uint32_t> meta;               // kernels.hpp bit layout
    View<int16_t> blen;                // -1 absorbed, else block length (valid for lists without ties)
    View<uint32_t> node_start;         // [n_nodes+1]
};

// Uninitialised, grow-only byte buffer (std::vector would zero-fill hundreds of MB per run).
struct ByteBuf {
    uint8_t *p = nullptr;
    size_t n = 0, cap = 0;
    size_t mapped = 0;                   // > 0: p is a mapping of that many bytes (large buffers: anonymous, huge pages asked for; or the file's)
    int fd = -1;                         // >= 0: the buffer IS a file (open_file): p is a shared mapping of it, what is written is the file's content
    bool reserve_uninit(size_t c);       // false: the host is out of memory (the buffer is then empty, p == nullptr).  A file-backed buffer keeps its content.
    bool grow_keep(size_t c);            // room for c bytes, the first n bytes kept
    // The buffer becomes the file `path` (created / truncated): a table assembled in it needs no fwrite afterwards -- on the GPU
    // boxes an 894 MB fwrite takes 116 ms, copying into already touched pages of a shared mapping 30 ms and its first touch, which
    // can be done beside the kernels, 90 ms (profiles/io_probe.cpp).  close_file() cuts the file to n bytes and closes it; the
    // buffer is empty afterwards.  Both return false on an I/O error.
    bool open_file(const char *path);
    bool close_file();
    const uint8_t *data() const { return p; }
    size_t size() const { return n; }
    ~ByteBuf();
    ByteBuf() {}
    ByteBuf(const ByteBuf &) = delete;
    ByteBuf &operator=(const ByteBuf &) = delete;
};

struct FinishStats { uint64_t exact_nodes = 0, inexact_nodes = 0, dirty_nodes = 0, big_nodes = 0; };

// packed: the 2-bit genome (needed only when a list has tied targets); may be empty if none has.
// n_threads host threads each serialise a contiguous range of source positions.
// Both return false when a host buffer could not be allocated (nothing is written through a null pointer).
bool serialize_inexact(const HostGenome &g, const InexactDeviceResult &r, View<uint64_t> packed,
                       ByteBuf &out, FinishStats &st, int n_threads, std::vector<size_t> *marks = nullptr);
// Splices the host-finished lists into the byte stream the device wrote (stages.hip, st_items_finish).
// arrival (optional): the stream is still on its way from the device; wait(c) returns once its first upto[c] bytes are in
// host memory (upto ascending, the last one = stream.size()).  The host lists are serialised first -- that needs none of
// the stream -- and each part of the stream is moved to its place as soon as it is there.
struct StreamArrival {
    std::vector<size_t> upto;
    std::function<bool(size_t)> wait;      // false: the copy failed
};
bool assemble_inexact(const HostGenome &g, View<uint8_t> stream, View<HostNode> hnodes, const InexactDeviceResult &host_items,
                      View<uint64_t> packed, uint64_t device_nodes, ByteBuf &out, FinishStats &st, int n_threads,
                      const StreamArrival *arrival = nullptr);

}  // namespace cora
