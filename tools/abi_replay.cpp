// abi_replay — runs a recorded sequence of C-ABI calls (include/catgan.h) with NO interpreter and NO PyTorch in the
// process: device memory from cg_malloc, one stream from cg_stream_create, every call dispatched by name through
// tools/abi_dispatch.inc (generated from the header).  It stands in for the LuaJIT host the build image cannot run:
// tests/test_abi_step.py records one whole D+G update (adversarial.lua:51-275) at the benchmarked batch - the description of the
// two networks (cg_net_create / _add / _bind ...), then per pass ONE cg_net_forward / cg_net_backward call with the criterion,
// batch assembly and optimiser calls between them, i.e. the PLANNED (fused) path the bench times - replays it here and requires
// the resulting parameter vectors to equal the Python host's bit for bit.  The same test replays the per-module walk
// (one C call per nn.Module method, the sequence the per-module Lua classes issue) at a small batch.
//
// Trace format (text, one record per line, fields separated by '|'):
//   seg|<index>|<bytes>|<offset of its initial contents in the blob file, or -1: start zeroed>
//   call|<entry point>|<arg>|<arg>...     arg = s (the stream) | n (NULL) | p:<seg>:<byte offset> | i:<int> | u:<uint> |
//                                               f:<hex float> | a:<k>:<p or n>,<p or n>,... (array of k device pointers)
//                                               cg_net_* / cg_graph_* (the planned executor): h:<k> handle k | H:<k> create handle k | o out-parameter |
//                                               L:<int>,<int>.. host long array | F:<hexfloat>,.. host float array | c:<text> |
//                                               v:<call index>:<byte offset> = inside the tensor that call returned
//   dump|p:<seg>:<offset>|<bytes>|<output file>
// usage: abi_replay <trace file> <blob file>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <fstream>
#include <sstream>
#include <string>
#include <vector>

#include "../include/catgan.h"

namespace {

std::vector<char*> g_seg;      // device base of every segment
void* g_stream = nullptr;
std::vector<void*> g_handles;  // cg_net / graph handles by index
std::vector<char*> g_vals;     // tensor address returned by call <index> (cg_net_forward / cg_net_backward)

[[noreturn]] void die(const std::string& m) { fprintf(stderr, "abi_replay: %s\n", m.c_str()); exit(2); }

void* resolve(const std::string& tok) {   // "n" | "p:<seg>:<off>"
    if (tok == "n") return nullptr;
    if (tok.size() >= 3 && tok[0] == 'v') {
        const size_t c = tok.find(':', 2);
        const size_t k = strtoull(tok.substr(2, c - 2).c_str(), nullptr, 10);
        if (k >= g_vals.size() || !g_vals[k]) die("value of a call that returned no tensor: " + tok);
        return g_vals[k] + atoll(tok.substr(c + 1).c_str());
    }
    if (tok.size() < 3 || tok[0] != 'p') die("bad pointer token '" + tok + "'");
    const size_t c = tok.find(':', 2);
    const long seg = atol(tok.substr(2, c - 2).c_str());
    const long long off = atoll(tok.substr(c + 1).c_str());
    if (seg < 0 || seg >= (long)g_seg.size() || !g_seg[seg]) die("pointer into unknown segment: " + tok);
    return g_seg[seg] + off;
}

struct Arg {
    std::string tok;
    std::vector<void*> arr;
    void* ptr() const { return tok == "s" ? g_stream : resolve(tok); }
    long long i() const { return atoll(tok.c_str() + 2); }
    unsigned long long u() const { return strtoull(tok.c_str() + 2, nullptr, 10); }
    double f() const { return strtod(tok.c_str() + 2, nullptr); }   // hex float: exact
    void* const* parr() const { return tok == "n" ? nullptr : arr.data(); }
    // planned-executor arguments
    mutable std::vector<long> la; mutable std::vector<float> fa; mutable long long scratch[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    void* handle() const { const size_t k = strtoull(tok.c_str() + 2, nullptr, 10); if (k >= g_handles.size()) die("unknown handle " + tok); return g_handles[k]; }
    void** newh() const { const size_t k = strtoull(tok.c_str() + 2, nullptr, 10); if (g_handles.size() <= k) g_handles.resize(k + 1, nullptr); return &g_handles[k]; }
    void* out() const { return (void*)scratch; }
    const long* larr() const { la.clear(); std::stringstream ss(tok.substr(2)); std::string it; while (std::getline(ss, it, ',')) if (!it.empty()) la.push_back(atol(it.c_str())); la.push_back(0); return la.data(); }
    const float* farr() const { fa.clear(); std::stringstream ss(tok.substr(2)); std::string it; while (std::getline(ss, it, ',')) if (!it.empty()) fa.push_back((float)strtod(it.c_str(), nullptr)); fa.push_back(0.f); return fa.data(); }
    const char* str() const { return tok.c_str() + 2; }
};

void need(const std::vector<Arg>& A, size_t n, const char* name) {
    if (A.size() != n) die(std::string(name) + ": recorded with " + std::to_string(A.size()) + " arguments, the header declares " + std::to_string(n));
}

int dispatch(const std::string& name, const std::vector<Arg>& A) {
#include "abi_dispatch.inc"
    die("no dispatch entry for " + name);
}

std::vector<std::string> split(const std::string& s, char sep) {
    std::vector<std::string> out;
    std::stringstream ss(s);
    std::string item;
    while (std::getline(ss, item, sep)) out.push_back(item);
    return out;
}

void ck(int rc, const std::string& what) {
    if (rc != 0) die(what + ": " + cg_last_error());
}

}  // namespace

int main(int argc, char** argv) {
    if (argc < 3) die("usage: abi_replay <trace> <blob>");
    std::ifstream tr(argv[1]);
    if (!tr) die(std::string("cannot open ") + argv[1]);
    FILE* blob = fopen(argv[2], "rb");
    if (!blob) die(std::string("cannot open ") + argv[2]);
    ck(cg_set_device(0), "cg_set_device");
    ck(cg_stream_create(&g_stream), "cg_stream_create");
    std::string line;
    long ncalls = 0;
    std::vector<char> host;
    while (std::getline(tr, line)) {
        if (line.empty()) continue;
        std::vector<std::string> f = split(line, '|');
        if (f[0] == "seg") {
            const size_t idx = strtoull(f[1].c_str(), nullptr, 10), bytes = strtoull(f[2].c_str(), nullptr, 10);
            const long long off = atoll(f[3].c_str());
            if (g_seg.size() <= idx) g_seg.resize(idx + 1, nullptr);
            void* d = nullptr;
            ck(cg_malloc(&d, bytes), "cg_malloc");
            g_seg[idx] = (char*)d;
            if (off >= 0) {
                host.resize(bytes);
                if (fseek(blob, off, SEEK_SET) != 0 || fread(host.data(), 1, bytes, blob) != bytes) die("short read in the blob file");
                ck(cg_memcpy_h2d(g_stream, d, host.data(), bytes), "cg_memcpy_h2d");
                ck(cg_stream_sync(g_stream), "cg_stream_sync");   // the host buffer is borrowed until the copy has run
            } else {
                ck(cg_memset_zero(g_stream, d, bytes), "cg_memset_zero");
            }
        } else if (f[0] == "call") {
            std::vector<Arg> A(f.size() - 2);
            for (size_t k = 2; k < f.size(); ++k) {
                Arg& a = A[k - 2];
                a.tok = f[k];
                if (a.tok[0] == 'a') {   // a:<k>:tok,tok,...
                    const size_t c = a.tok.find(':', 2);
                    for (const std::string& t : split(a.tok.substr(c + 1), ',')) a.arr.push_back(resolve(t));
                }
            }
            ck(dispatch(f[1], A), f[1]);
            if (f[1] == "cg_net_forward" || f[1] == "cg_net_backward" || f[1] == "cg_net_forward_pair" || f[1] == "cg_net_pair_join") {
                // keep the returned tensor's address under this call's index (cg_net_forward_pair: the first pass's, cg_net_pair_join: the second's)
                const size_t yi = f[1] == "cg_net_forward" ? 10 : (f[1] == "cg_net_backward" ? 7 : (f[1] == "cg_net_forward_pair" ? 14 : 2));
                if ((size_t)ncalls >= g_vals.size()) g_vals.resize(ncalls + 1, nullptr);
                g_vals[ncalls] = *(char**)A[yi].out();
            }
            ++ncalls;
        } else if (f[0] == "dump") {
            ck(cg_stream_sync(g_stream), "cg_stream_sync");
            const size_t bytes = strtoull(f[2].c_str(), nullptr, 10);
            host.resize(bytes);
            ck(cg_memcpy_d2h(g_stream, host.data(), resolve(f[1]), bytes), "cg_memcpy_d2h");
            ck(cg_stream_sync(g_stream), "cg_stream_sync");
            FILE* o = fopen(f[3].c_str(), "wb");
            if (!o || fwrite(host.data(), 1, bytes, o) != bytes) die("cannot write " + f[3]);
            fclose(o);
        } else {
            die("unknown record '" + f[0] + "'");
        }
    }
    ck(cg_stream_sync(g_stream), "cg_stream_sync");
    for (char* p : g_seg)
        if (p) cg_free(p);
    cg_stream_destroy(g_stream);
    fclose(blob);
    printf("abi_replay: %ld calls replayed through %zu segments\n", ncalls, g_seg.size());
    return 0;
}
