// hip_forward_pipe.h -- the MI355X backend behind Sayuri's NetworkForwardPipe plugin interface.
//
// Counterpart of CudaForwardPipe (reference src/neural/cuda/cuda_forward_pipe.{h,cc}) +
// BatchForwardPipe (reference src/neural/batch_forward_pipe.{h,cc}):
//   * same seven entry points (Initialize / Forward / Construct / Release / Destroy / Valid /
//     GetNumWorkers) plus BatchForward(gpu, inputs), same argument meaning, same error
//     behaviour (device/API failures surface as std::runtime_error, cuda_common.cc:46-62);
//   * the device side is reached only through the C-ABI of include/sayuri_hip.h;
//   * the leaf-batch collector is rewritten: callers write their planes ONCE, straight into a
//     pinned staging slot of the batch being assembled (the reference copies the 62 KB
//     InputData three times per evaluation: batch_forward_pipe.cc:9, :178,
//     cuda_forward_pipe.cc:696-701), one persistent pump thread per GPU owns its ctx and its
//     own queue shard, and completion is a per-request flag instead of a heap-allocated
//     mutex+condvar pair per leaf.
#pragma once

#include <atomic>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#ifdef SAYURI_IN_TREE
// Built inside the reference tree: the reference's own plugin types (global namespace).
#include "neural/description.h"
#include "neural/network_basic.h"
#include "packed_planes.h"
using sayuri_host::PackedPlanes;
#define SAYURI_HOST_BEGIN
#define SAYURI_HOST_END
#define SAYURI_EXT_OVERRIDE  // the reference's NetworkForwardPipe has no packed entry points: plain members there
#else
#include "packed_planes.h"
#include "pipe_api.h"
#include "weights_model.h"
#define SAYURI_HOST_BEGIN namespace sayuri_host {
#define SAYURI_HOST_END }
#define SAYURI_EXT_OVERRIDE override
#endif

struct sayuri_hip_ctx;

SAYURI_HOST_BEGIN

// What the reference reads from its global option map (config.cc:21-133): "batch_size",
// "gpus", "fp16", "gpu_waittime", "defualt_boardsize", "fixed_nn_boardsize".
struct HipPipeConfig {
    int batch_size{256};
    std::vector<int> gpus;       // empty = every visible device
    bool fp16{true};
    int gpu_waittime_ms{2};
    int default_boardsize{kBoardSize};
    int fixed_nn_boardsize{0};
};

class HipForwardPipe : public NetworkForwardPipe {
public:
    explicit HipForwardPipe(HipPipeConfig cfg = {});
    ~HipForwardPipe() override;

    void Initialize(std::shared_ptr<DNNWeights> weights) override;
    OutputResult Forward(const InputData& input) override;
    void Construct(ForwardPipeOption option, std::shared_ptr<DNNWeights> weights) override;
    void Release() override;
    void Destroy() override;
    bool Valid() const override;
    int GetNumWorkers() const override { return static_cast<int>(graphs_.size()); }

    // The compact-input flavour of Forward (packed_planes.h): same collector, same result
    // bit for bit, 1.8 KB of bit planes per request through staging and PCIe instead of 62 KB of fp32 planes.  A batch
    // made of packed requests only goes to the GPU packed (sayuri_hip_submit_packed); in a mixed batch the pump expands
    // the packed ones into the fp32 staging first.
    OutputResult ForwardPacked(const PackedPlanes& input) SAYURI_EXT_OVERRIDE;
    bool AcceptsPacked() const SAYURI_EXT_OVERRIDE { return true; }

    // batch_forward_pipe.h:27-28.  `inputs` are already re-padded into the NN grid.
    std::vector<OutputResult> BatchForward(int gpu, const std::vector<InputData>& inputs);

    // Non-blocking flavour for M:N self-play schedulers: the result lands in *out and
    // *done flips to 1 (release) once the batch containing it has been evaluated.
    void Submit(const InputData& input, OutputResult* out, std::atomic<int>* done);
    // (Submit: the pump fills *out and flips *done.  Forward: the caller is woken through the batch's wake tree and
    // copies its own result.)

    int board_size() const { return board_size_; }
    int max_batch() const { return max_batch_; }
    bool fp16() const { return cfg_.fp16; }
    sayuri_hip_ctx* ctx(int gpu) const;
    size_t num_batches() const { return batches_.load(std::memory_order_relaxed); }
    size_t num_evals() const { return evals_.load(std::memory_order_relaxed); }
    // pump time split since construction, microseconds: [0] inside sayuri_hip_forward, [1] filling
    // outputs + signalling, [2] waiting for a batch to form, [3] waiting for plane copies, [4] waking callers parked
    // for a staging set, [5] batches sent with fewer than batch_size requests (a count, not a time)
    // [6] time the GPU queue was empty while the pipe was in use, [7] waiting for callers to finish their plane copies
    void pump_times(double out[8]) const {
        for (int i = 0; i < 8; ++i) out[i] = static_cast<double>(pump_ns_[i].load(std::memory_order_relaxed)) * 1e-3;
    }

private:
    struct Echo {  // what FillOutput needs of the request once its planes are staged
        int board_size;
        PolicyBufferOffset offset;
        float komi;
    };
    struct Request {
        Echo echo;
        OutputResult* output;
        std::atomic<int>* done;
        bool self_serve;  // a blocking Forward() caller: woken through the tree, takes its result itself
        bool fiber;       // a Forward() caller that is a fiber (fiber.h): takes its result itself, no futex: the pump only
                          // flips `done`, the fiber's scheduler thread sees it
    };
    // One batch being assembled or evaluated.  Callers reserve a slot with one atomic add and copy
    // their planes straight into the pinned buffer the GPU will read (one copy per evaluation,
    // done in parallel by the calling threads); `ready` counts finished copies.
    struct Staging {
        static constexpr unsigned kClosed = 1u << 31;
        float* planes{nullptr};
        std::uint32_t* packed{nullptr};      // packed records of the slots filled through ForwardPacked (pinned)
        std::vector<std::uint8_t> is_packed;  // per slot: its planes are in `packed`, not in `planes`
        float *prob{nullptr}, *pass{nullptr}, *misc{nullptr}, *own{nullptr};
        std::vector<int> bsz;
        std::vector<Request> reqs;
        std::atomic<unsigned> reserved{0};  // slot counter | kClosed
        std::atomic<unsigned> ready{0};
        int n_inflight{0};   // pump-private: batch size while on the GPU
        int ticket{-1};      // pump-private: sayuri_hip_submit ticket
        // ---- hand-out of a finished batch to blocking Forward() callers.  The pump wakes ONE of them; every woken
        // caller wakes up to kFanout others (a tree over the batch, log depth), then copies its own result out of the
        // pinned output buffers.  Waking a sleeping thread costs microseconds of kernel time: done serially by the pump
        // it is 0.5 ms for 256 requests and 8 ms for 1024; spread over the callers it is a few wake-ups each, in
        // parallel.  The fin_* arrays are a snapshot of the batch, so the set can take new requests at once.
        static constexpr int kFanout = 8;
        std::vector<Request> fin_reqs;      // requests of the finished batch, blocking callers first come in fin_list order
        std::vector<int> fin_list;          // slots of the blocking callers
        std::vector<int> fin_pos;           // slot -> position in fin_list
        int fin_count{0};
        int fin_fibers{0};                  // fiber callers of the finished batch (they also count in `consumed`)
        std::atomic<int> fin_status{0};     // 1 ok, -1 failed
        std::atomic<int> wakes_done{0};     // callers that have woken their children
        std::atomic<int> consumed{0};       // callers that have taken their result
    };
    struct Graph {  // one per GPU (NNGraph in the reference)
        int device{-1};
        sayuri_hip_ctx* ctx{nullptr};
        static constexpr int kSets = 4;  // ring of staging sets: one fills, up to two are on the GPU, the rest are
                                         // full and queued -- with more leaves in flight than two batches the next
                                         // batch is always ready when the GPU finishes one
        Staging st[kSets];
        std::atomic<int> fill{0};    // index of the staging set new requests go to
        std::atomic<int> epoch{0};   // bumped whenever a set re-opens or the fill index moves: callers that found
                                     // both sets taken sleep on it (futex) instead of polling
        std::thread pump;
        std::mutex dev_mu;  // owner of ctx (pump batch or a direct BatchForward)
        std::mutex mu;      // pump sleep / wake-up only
        std::condition_variable cv;
    };

    struct Ticket { Graph* g; Staging* s; int slot; };
    // exactly one of input / packed is non-null
    Ticket Reserve(const InputData* input, const PackedPlanes* packed, OutputResult* out, std::atomic<int>* done, bool self_serve,
                   bool fiber = false);
    OutputResult ForwardAny(const InputData* input, const PackedPlanes* packed);
    int BinaryPlanes() const;
    void Reopen(Graph* g, Staging* s);
    void BuildGraphs();
    void DestroyGraphs();
    void PumpLoop(Graph* g);
    void SubmitBatch(Graph* g, Staging* s, int n);
    void FinishBatch(Graph* g, Staging* s, int n);
    void StageInput(Staging* s, int slot, const InputData& in, bool already_padded);
    void StagePacked(Staging* s, int slot, const PackedPlanes& in);
    void ExpandPacked(Staging* s, int slot);
    void FillOutput(const Staging* s, int slot, const Echo& in, bool unpad, OutputResult* out) const;

    HipPipeConfig cfg_;
    int board_size_{0};
    int max_batch_{0};
    std::atomic<int> forward_size_{0};  // batch size the collector forwards at (<= max_batch_); Construct() may lower it live
    std::vector<std::unique_ptr<Graph>> graphs_;
    std::atomic<bool> running_{false};
    std::atomic<bool> fibers_seen_{false};  // some caller is a fiber: re-opened staging sets are announced to the fiber schedulers too
    std::atomic<unsigned> next_graph_{0};
    std::atomic<size_t> batches_{0}, evals_{0};
    mutable std::atomic<long long> pump_ns_[8] = {};
    // SAYURI_PIPE_TRACE=1 (read once at construction; a measuring aid, printed to stderr when the graphs are destroyed):
    // when do requests arrive relative to the last finished batch, how full are the batches and why were they closed
    bool trace_{false};
    bool rotate_notify_{true};  // SAYURI_AB_ROTATE_NOTIFY (measuring aid)
    std::atomic<int> epoch_parked_{0};  // fibers suspended in Reserve() until a staging set re-opens
    std::atomic<long long> trace_last_done_ns_{0};
    std::atomic<long> trace_arrival_[32] = {};  // 250 us bins since the last finished batch
    std::atomic<long> trace_size_[17] = {};     // batch size / 16
    std::atomic<long> trace_resume_[32] = {};   // fibers: batch finished -> the game runs again
    std::atomic<long> trace_think_[32] = {};    // fibers: the game runs again -> its next request
    std::atomic<long> trace_tail_[32] = {};     // tail rule: how long the running batch had been running when the set was closed
    std::atomic<long> trace_reason_[4] = {};
    double tail_frac_{0.93};                    // SAYURI_PIPE_TAIL (measuring aid): the fraction of a batch's time after which a partial set is enqueued behind it    // closed because: full, GPU idle + wait expired, 93 % of the running batch, stray
    void TraceDump() const;
};

SAYURI_HOST_END
