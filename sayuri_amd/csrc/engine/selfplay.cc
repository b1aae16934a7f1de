#include "selfplay.h"

#include "fiber.h"

#include <dirent.h>
#include <sys/stat.h>

#include <malloc.h>
#include <pthread.h>
#include <sched.h>
#include <zlib.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <ctime>
#include <fstream>
#include <random>
#include <sstream>
#include <stdexcept>

namespace sayuri_engine {

using sayuri_go::kAreaScoring;
using sayuri_go::kBlack;
using sayuri_go::kPassMove;
using sayuri_go::kTerritoryScoring;
using sayuri_go::kWhite;

namespace {
bool IsZeroKomi(float v) { return std::abs(v) < 1e-4f; }

std::vector<std::string> SplitColon(std::string s) {
    for (char& c : s)
        if (c == ':') c = ' ';
    std::istringstream in(s);
    std::vector<std::string> out;
    for (std::string w; in >> w;) out.push_back(w);
    return out;
}

bool DirExists(const std::string& d) {
    struct stat st;
    return stat(d.c_str(), &st) == 0 && S_ISDIR(st.st_mode);
}
void MakeDir(const std::string& d) {
    if (!DirExists(d)) mkdir(d.c_str(), 0755);
}
std::string NowString() {
    char buf[64];
    const std::time_t t = std::time(nullptr);
    std::tm tm_buf;
    localtime_r(&t, &tm_buf);  // std::localtime shares one static buffer between the game threads
    std::strftime(buf, sizeof(buf), "%Y-%m-%d-%H:%M:%S", &tm_buf);
    return buf;
}
} // namespace

float AdjustKomiToHalf(float komi) {
    // round to the nearest half point, keeping the sign
    if (IsZeroKomi(komi)) return 0;
    const bool negative = komi < 0.0f;
    if (negative) komi = -komi;
    const int whole = static_cast<int>(komi);
    float frac = komi - whole;
    frac = frac < 0.25f ? 0.f : frac < 0.75f ? 0.5f : 1.f;
    komi = frac + whole;
    if (negative && !IsZeroKomi(komi)) komi = -komi;
    return komi;
}

// ---------------------------------------------------------------------------------------------
SelfplayEngine::SelfplayEngine(std::shared_ptr<NetworkForwardPipe> pipe, int weights_version, const EngineOptions& opt)
    : opt_(opt) {
    network_.Initialize(std::move(pipe), weights_version, opt_.network);
    const int games = std::max(1, opt_.selfplay.parallel_games);
    std::uint64_t seed = opt_.selfplay.seed;
    if (seed == 0) seed = static_cast<std::uint64_t>(std::chrono::steady_clock::now().time_since_epoch().count());
    for (int g = 0; g < games; ++g) {
        auto slot = std::make_unique<Slot>();
        slot->state.Reset(opt_.selfplay.default_boardsize, opt_.selfplay.default_komi, opt_.selfplay.scoring_rule);
        slot->search = std::make_unique<Search>(slot->state, network_, opt_.search);
        // two independent streams per game, all derived from one seed
        slot->search->Seed(seed + 2 * static_cast<std::uint64_t>(g) + 1, seed + 2 * static_cast<std::uint64_t>(g) + 2);
        slots_.push_back(std::move(slot));
    }
    ParseQueries();
}

SelfplayEngine::Slot& SelfplayEngine::At(int g) {
    if (g < 0 || g >= GetParallelGames()) throw std::runtime_error("The game index is out of array.");
    return *slots_[static_cast<size_t>(g)];
}

std::uint64_t SelfplayEngine::playouts() const {
    std::uint64_t n = 0;
    for (const auto& s : slots_) n += s->search->total_playouts();
    return n;
}

void SelfplayEngine::ParseQueries() {
    float total = 0.f;
    for (const auto& q : opt_.selfplay.selfplay_queries) {
        const auto w = SplitColon(q);
        if (w.empty()) break;
        if (w[0] == "bkp" && w.size() == 4) { // board : komi : probability
            BoardQuery b{std::stoi(w[1]), std::stof(w[2]), std::stof(w[3])};
            board_queries_.push_back(b);
            total += b.prob;
        } else if (w[0] == "bhp" && w.size() == 4) { // board : max handicap : probability
            HandicapQuery h{std::stoi(w[1]), std::stoi(w[2]), std::stof(w[3])};
            if (h.handicaps >= 2) handicap_queries_.push_back(h);
        } else if (w[0] == "srs") { // scoring rule set
            for (size_t i = 1; i < w.size(); ++i) scoring_set_.push_back(w[i] == "territory" ? kTerritoryScoring : kAreaScoring);
        }
    }
    if (board_queries_.empty()) {
        board_queries_.push_back({opt_.selfplay.default_boardsize, opt_.selfplay.default_komi, 1.f});
    } else {
        for (auto& q : board_queries_) q.prob /= total;
    }
    if (scoring_set_.empty()) scoring_set_.push_back(kAreaScoring);
    const bool has_territory = std::find(scoring_set_.begin(), scoring_set_.end(), kTerritoryScoring) != scoring_set_.end();
    const bool has_area = std::find(scoring_set_.begin(), scoring_set_.end(), kAreaScoring) != scoring_set_.end();
    if (has_territory && !has_area) scoring_set_.push_back(kTerritoryScoring); // as the reference does (engine.cc:163-168)
    std::sort(scoring_set_.begin(), scoring_set_.end());
    scoring_set_.erase(std::unique(scoring_set_.begin(), scoring_set_.end()), scoring_set_.end());
}

void SelfplayEngine::PrepareGame(int g) {
    Slot& s = At(g);
    Rng& rng = s.search->caller_rng();
    s.state.ClearBoard();
    s.state.SetRule(kAreaScoring);
    s.comments.clear();

    constexpr std::uint32_t kRange = 1000000;
    const std::uint32_t pick = rng.Below(kRange);
    float acc = 0.f;
    size_t chosen = 0;
    for (size_t i = 0; i < board_queries_.size(); ++i) {
        acc += board_queries_[i].prob;
        if (pick <= kRange * acc) {
            chosen = i;
            break;
        }
    }
    auto rules = scoring_set_;
    std::shuffle(rules.begin(), rules.end(), rng);
    s.state.Reset(board_queries_[chosen].board_size, board_queries_[chosen].komi, rules.front());
    const int h = GetHandicaps(g);
    if (h > 0) SetHandicapGame(g, h);
    else SetNormalGame(g);
}

int SelfplayEngine::GetHandicaps(int g) {
    Slot& s = At(g);
    Rng& rng = s.search->caller_rng();
    for (const auto& q : handicap_queries_) {
        if (s.state.GetBoardSize() == q.board_size && rng.Chance(q.prob))
            return static_cast<int>(rng.Next() % static_cast<std::uint64_t>(q.handicaps - 1)) + 2;
    }
    return 0;
}

void SelfplayEngine::SetNormalGame(int g) {
    if (At(g).search->caller_rng().Chance(opt_.selfplay.random_opening_prob)) SetRandomOpeningGame(g);
    SetUnfairKomi(g);
}

void SelfplayEngine::SetHandicapGame(int g, int handicaps) {
    Slot& s = At(g);
    Rng& rng = s.search->caller_rng();
    // handicap stones go where the policy would put them
    for (int i = 0; i < handicaps - 1; ++i) {
        s.state.SetToMove(kBlack);
        const int move = network_.GetVertexWithPolicy(s.state, 0.8f, false, rng);
        s.state.AppendMove(move, kBlack);
    }
    s.state.SetHandicap(handicaps);
    SetFairKomi(g);
    if (rng.Chance(opt_.selfplay.random_opening_prob)) SetRandomOpeningGame(g);
    if (!rng.Chance(opt_.selfplay.handicap_fair_komi_prob)) SetUnfairKomi(g);
}

void SelfplayEngine::SetRandomOpeningGame(int g) {
    Slot& s = At(g);
    Rng& rng = s.search->caller_rng();
    const int bs = s.state.GetBoardSize();
    const int random_moves = static_cast<int>(opt_.search.random_moves_factor * s.state.GetNumIntersections());
    std::normal_distribution<float> dist(0.f, static_cast<float>(bs) / 4);
    const int remaining = std::max(static_cast<int>(dist(rng)) + random_moves - s.state.GetMoveNumber(), 0);
    const float lambda = 0.69314718056f / bs;
    int times = 0;
    for (int i = 0; i < remaining; ++i) {
        if (s.state.GetPasses() >= 2) break;
        const float temp = std::max(opt_.selfplay.random_opening_temp * std::exp(-(lambda * times)), 0.8f);
        s.state.PlayMove(network_.GetVertexWithPolicy(s.state, temp, false, rng));
        s.comments.emplace_back();
        times += 1;
    }
    SetFairKomi(g);
}

void SelfplayEngine::SetUnfairKomi(int g) {
    Slot& s = At(g);
    Rng& rng = s.search->caller_rng();
    float stddev = opt_.selfplay.komi_stddev;
    if (rng.Chance(opt_.selfplay.komi_big_stddev_prob)) stddev = opt_.selfplay.komi_big_stddev;
    std::normal_distribution<float> dist(0.f, stddev);
    const float bonus = dist(rng);
    s.state.SetKomi(AdjustKomiToHalf(s.state.GetKomi() + bonus));
}

void SelfplayEngine::SetFairKomi(int g) {
    Slot& s = At(g);
    const auto result = s.search->Computation(opt_.search.playouts, Search::kNoExploring);
    float lead = result.root_score_lead;
    if (s.state.GetToMove() == kWhite) lead = 0.0f - lead;
    s.state.SetKomi(AdjustKomiToHalf(s.state.GetKomi() + lead));
}

void SelfplayEngine::PlayPolicyMoves(int g, int count) {
    Slot& s = At(g);
    Rng& rng = s.search->caller_rng();
    for (int i = 0; i < count; ++i) {
        if (s.state.GetPasses() >= 2 || s.state.IsGameOver()) break;
        s.state.PlayMove(network_.GetVertexWithPolicy(s.state, 1.0f, false, rng));
        s.comments.emplace_back();
    }
}

std::string SelfplayEngine::SelectWeights() const {
    if (!opt_.selfplay.weights_file.empty() && opt_.selfplay.weights_dir.empty()) return opt_.selfplay.weights_file;
    std::string best = opt_.selfplay.weights_file;
    if (opt_.selfplay.weights_dir.empty()) return best;
    DIR* d = opendir(opt_.selfplay.weights_dir.c_str());
    if (!d) return best;
    bool have = false;
    struct timespec best_time {};
    while (struct dirent* e = readdir(d)) {
        const std::string name = e->d_name;
        if (name == "." || name == "..") continue;
        const std::string path = opt_.selfplay.weights_dir + "/" + name;
        struct stat st {};
        if (stat(path.c_str(), &st) != 0 || !S_ISREG(st.st_mode)) continue;
        const bool newer = !have || st.st_mtim.tv_sec > best_time.tv_sec ||
                           (st.st_mtim.tv_sec == best_time.tv_sec && st.st_mtim.tv_nsec > best_time.tv_nsec);
        if (newer) { best = path; best_time = st.st_mtim; have = true; }
    }
    closedir(d);
    return best;
}

bool SelfplayEngine::ShouldHalt() const {
    if (opt_.selfplay.weights_dir.empty()) return false;
    const std::string newest = SelectWeights();
    if (newest == opt_.selfplay.weights_file) return false;
    // two spellings of one file (trailing slash in weights_dir, relative vs absolute path, a symlink) are not a newer
    // network: compare the files, not the strings
    struct stat a, b;
    if (stat(newest.c_str(), &a) == 0 && stat(opt_.selfplay.weights_file.c_str(), &b) == 0 && a.st_dev == b.st_dev && a.st_ino == b.st_ino)
        return false;
    return true;
}

bool SelfplayEngine::Step(int g) {
    Slot& s = At(g);
    if (s.state.IsGameOver()) return false;
    int move = s.search->GetSelfPlayMove();
    if (move_cap > 0 && s.state.GetMoveNumber() >= move_cap) move = kPassMove;
    s.state.PlayMove(move);
    s.comments.resize(static_cast<size_t>(s.state.GetMoveNumber()) + 1);
    s.comments[static_cast<size_t>(s.state.GetMoveNumber())] = s.search->last_comment();
    moves_.fetch_add(1, std::memory_order_relaxed);
    return !s.state.IsGameOver();
}

void SelfplayEngine::Selfplay(int g) {
    while (Step(g)) {
    }
    At(g).search->UpdateTerritoryHelper();
}

void SelfplayEngine::GatherTrainingData(std::vector<TrainingData>& chunk, int g) { At(g).search->GatherTrainingBuffer(chunk); }

std::string SelfplayEngine::GatherSgfString(int g) {
    Slot& s = At(g);
    GameState& st = s.state;
    std::ostringstream out;
    const std::string bot = "sayuri-amd 0.1";
    out << "(;GM[1]FF[4]SZ[" << st.GetBoardSize() << "]KM[" << st.GetKomi() << "]RU["
        << (st.GetScoringRule() == kAreaScoring ? "chinese" : "japanese") << "]PB[" << bot << "]PW[" << bot << "]DT[" << NowString() << ']';
    if (st.GetHandicap() != 0) out << "HA[" << st.GetHandicap() << ']';
    for (int c = 0; c < 2; ++c) {
        const auto setup = st.GetAppendMoves(c);
        if (setup.empty()) continue;
        out << (c == kBlack ? "AB" : "AW");
        for (int v : setup) out << '[' << st.VertexToSgf(v) << ']';
    }
    out << "C[" << (st.GetScoringRule() == kAreaScoring ? "chinese" : "japanese") << ']';
    const float score = st.GetFinalScore(kBlack);
    const bool pass_end = st.GetPasses() >= 2;
    if (pass_end) st.SetWinner(score > 1e-4 ? sayuri_go::kBlackWon : score < -1e-4 ? sayuri_go::kWhiteWon : sayuri_go::kDrawGame);
    if (st.GetWinner() != sayuri_go::kUndecided) {
        out << "RE[";
        if (st.GetWinner() == sayuri_go::kBlackWon) {
            out << "B+";
            if (pass_end) out << score;
        } else if (st.GetWinner() == sayuri_go::kWhiteWon) {
            out << "W+";
            if (pass_end) out << -score;
        } else {
            out << "0";
        }
        if (!pass_end && st.GetWinner() != sayuri_go::kDrawGame) out << "Resign";
        out << ']';
    }
    for (int i = 1; i <= st.GetMoveNumber(); ++i) {
        const auto mv = st.MoveAt(i);
        out << ';' << (mv.second == kBlack ? 'B' : 'W') << '[' << st.VertexToSgf(mv.first) << ']';
        if (static_cast<size_t>(i) < s.comments.size() && !s.comments[static_cast<size_t>(i)].empty()) out << "C[" << s.comments[static_cast<size_t>(i)] << ']';
    }
    out << ')';
    return out.str();
}

// ---------------------------------------------------------------------------------------------
SelfplayPipe::SelfplayPipe(std::shared_ptr<NetworkForwardPipe> pipe, int weights_version, const EngineOptions& opt,
                           const std::string& name_suffix)
    : opt_(opt), engine_(std::move(pipe), weights_version, opt) {
    max_games_.store(std::max(opt_.selfplay.num_games, engine_.GetParallelGames()));
    const std::string& target = opt_.selfplay.target_directory;
    if (!target.empty()) {
        // a name not used by an earlier run in this directory (pipe.cc:44-80), plus the caller's suffix (rank id)
        for (;;) {
            std::ostringstream ss;
            const std::time_t now = std::time(nullptr);
            ss << std::hex << std::uppercase << std::hash<std::string>()(std::to_string(now) + "/" + std::to_string(opt_.selfplay.seed));
            hash_ = ss.str() + name_suffix;
            tdata_dir_ = target + "/tdata/" + hash_;
            vdata_dir_ = target + "/vdata/" + hash_;
            if (!DirExists(tdata_dir_) && !DirExists(vdata_dir_)) break;
            std::this_thread::sleep_for(std::chrono::seconds(1));
        }
        sgf_dir_ = target + "/sgf";
        queries_dir_ = target + "/net_queries";
        if (!DirExists(target)) throw std::runtime_error("ABORT: Target directory do not exist.");
        MakeDir(target + "/tdata");
        MakeDir(tdata_dir_);
        MakeDir(target + "/vdata");
        MakeDir(vdata_dir_);
        MakeDir(sgf_dir_);
        MakeDir(queries_dir_);
    }
}

bool SelfplayPipe::WriteGzip(const std::string& name, const std::string& text) {
    const std::string path = name + ".gz";
    gzFile f = gzopen(path.c_str(), "wb9");
    if (!f) return false;
    const int n = text.empty() ? 0 : gzwrite(f, text.data(), static_cast<unsigned>(text.size()));
    const bool closed = gzclose(f) == Z_OK;  // the flush of the last block happens here: a full disk shows up in this call
    if (!closed || !(text.empty() || n > 0)) return false;
    struct stat sb;
    if (::stat(path.c_str(), &sb) == 0) bytes_written_.fetch_add(static_cast<std::uint64_t>(sb.st_size), std::memory_order_relaxed);
    text_bytes_.fetch_add(text.size(), std::memory_order_relaxed);
    return true;
}

bool SelfplayPipe::SaveChunk(int id, float vdata_prob, std::vector<TrainingData>& chunk, Rng& rng) {
    std::ostringstream tdata, vdata;
    vdata_prob = std::max(std::min(vdata_prob, 1.0f), 0.0f);
    for (auto& d : chunk) {
        if (rng.Chance(1.0f - vdata_prob)) d.StreamOut(tdata);
        else d.StreamOut(vdata);
        if (!d.discard) records_.fetch_add(1, std::memory_order_relaxed);
    }
    bool ok = WriteGzip(tdata_dir_ + "/game_" + std::to_string(id) + ".txt", tdata.str());
    ok &= WriteGzip(vdata_dir_ + "/game_" + std::to_string(id) + ".txt", vdata.str());
    chunk.clear();
    return ok;
}

void SelfplayPipe::SaveSgf(const std::string& sgf) {
    std::ofstream f(sgf_dir_ + "/" + hash_ + ".sgf", std::ios_base::app);
    if (!f.is_open()) return;
    f << sgf << std::endl;
    if (f.good()) bytes_written_.fetch_add(sgf.size() + 1, std::memory_order_relaxed);
}

void SelfplayPipe::SaveNetQueries(int games, const std::string& text) {
    std::ofstream f(queries_dir_ + "/" + hash_ + ".txt", std::ios_base::app);
    if (!f.is_open()) return;
    f << games << " " << text << std::endl;
    if (f.good()) bytes_written_.fetch_add(text.size() + 8, std::memory_order_relaxed);
}

void SelfplayPipe::WriterLoop() {
    pthread_setname_np(pthread_self(), "sayuri-writer");
    // chunks leave in random order, one game per file; a pool of `games` finished games is kept while the
    // workers run so that consecutive files do not come from consecutive games (pipe.cc:181-232)
    constexpr float kValidationRatio = 0.1f;
    const int games = engine_.GetParallelGames();
    Rng rng(opt_.selfplay.seed ^ 0x5eedf00dULL);
    std::vector<std::shared_ptr<DataSgf>> pool;
    const int pool_games = opt_.selfplay.chunk_pool_games > 0 ? std::min(opt_.selfplay.chunk_pool_games, games) : games;
    auto publish_cpu = [this]() {
        struct timespec ts;
        if (clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts) == 0)
            writer_cpu_ns_.store(static_cast<std::uint64_t>(ts.tv_sec) * 1000000000ull + static_cast<std::uint64_t>(ts.tv_nsec), std::memory_order_relaxed);
    };
    bool keep = true;
    while (keep) {
        publish_cpu();
        std::this_thread::sleep_for(std::chrono::milliseconds(20));
        keep = writer_running_.load(std::memory_order_relaxed);
        std::deque<std::pair<int, std::string>> queries;
        {
            std::lock_guard<std::mutex> lock(data_mu_);
            while (!data_queue_.empty()) {
                pool.push_back(data_queue_.front());
                data_queue_.pop_front();
                keep = true;
            }
            queries.swap(queries_queue_);
        }
        const bool running = writer_running_.load(std::memory_order_relaxed);
        const size_t hold = running ? static_cast<size_t>(pool_games) : 1;
        if (!running && pool.size() > 8 && !opt_.selfplay.target_directory.empty()) {
            // The run is over and the pool is flushed (pipe.cc:206-208 drops the hold to 1): hundreds of games at gzip level 9,
            // 0.15 s each.  Every chunk is a file of its own, so the flush goes over a few threads -- ids, the shuffle and the
            // tdata / vdata split of every game are drawn here, in order, from the one generator; the SGF lines follow in id order.
            // The ids are handed out up front, so a chunk that fails to write (gzopen / gzwrite / gzclose) leaves a GAP in the
            // game_N numbering of this last flush and is not counted in chunks_ (the serial path below reuses the id; nothing is
            // numbered after this flush, so a gap never collides with a later chunk).
            std::shuffle(pool.begin(), pool.end(), rng);
            struct Job { std::shared_ptr<DataSgf> item; int id; std::uint64_t seed; bool ok; };
            std::vector<Job> jobs;
            int id = static_cast<int>(chunks_.load());
            while (!pool.empty()) {
                jobs.push_back(Job{pool.back(), id++, rng.Next(), false});
                pool.pop_back();
            }
            std::atomic<size_t> next{0};
            const unsigned nthreads = std::min<unsigned>(8u, std::max(1u, std::thread::hardware_concurrency()));
            std::vector<std::thread> helpers;
            std::atomic<std::uint64_t> helper_cpu_ns{0};
            auto work = [&]() {
                for (size_t k = next.fetch_add(1); k < jobs.size(); k = next.fetch_add(1)) {
                    Rng local(jobs[k].seed);
                    jobs[k].ok = SaveChunk(jobs[k].id, kValidationRatio, jobs[k].item->first, local);
                }
                struct timespec ts;
                if (clock_gettime(CLOCK_THREAD_CPUTIME_ID, &ts) == 0)
                    helper_cpu_ns.fetch_add(static_cast<std::uint64_t>(ts.tv_sec) * 1000000000ull + static_cast<std::uint64_t>(ts.tv_nsec));
            };
            for (unsigned t = 0; t < nthreads; ++t) helpers.emplace_back(work);
            for (auto& h : helpers) h.join();
            for (auto& j : jobs) {
                if (j.ok) chunks_.fetch_add(1);
                SaveSgf(j.item->second);
            }
            flush_helper_cpu_ns_ = helper_cpu_ns.load();
        }
        while (pool.size() >= hold && !pool.empty()) {
            std::shuffle(pool.begin(), pool.end(), rng);
            auto item = pool.back();
            pool.pop_back();
            if (!opt_.selfplay.target_directory.empty()) {
                if (SaveChunk(static_cast<int>(chunks_.load()), kValidationRatio, item->first, rng)) chunks_.fetch_add(1);
                SaveSgf(item->second);
                publish_cpu();
            } else {
                for (auto& d : item->first)
                    if (!d.discard) records_.fetch_add(1, std::memory_order_relaxed);
            }
        }
        if (!opt_.selfplay.target_directory.empty())
            for (auto& q : queries) SaveNetQueries(q.first, q.second);
    }
    publish_cpu();
}

void SelfplayPipe::WindDown() {
    // pipe.cc:248-254: finish the games in flight plus a buffer, rounded up to a multiple of 25
    constexpr int kBufferGames = 25;
    const int accum = std::max(engine_.GetParallelGames(), accumulation_games_.load(std::memory_order_relaxed));
    const int cap = (accum + kBufferGames + kBufferGames - 1) / kBufferGames * kBufferGames;
    int cur = max_games_.load();
    while (cap < cur && !max_games_.compare_exchange_weak(cur, cap)) {
    }
}

namespace {
// CPUs worth of time this process can get: the affinity mask, limited by a cgroup v2 / v1 CPU quota if there is one
int UsableCores(int affinity_cpus) {
    int cores = std::max(1, affinity_cpus);
    auto apply = [&cores](double quota, double period) {
        if (quota > 0 && period > 0) cores = std::max(1, std::min(cores, static_cast<int>(quota / period + 0.5)));
    };
    if (FILE* f = std::fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char q[64] = {0};
        double period = 0;
        if (std::fscanf(f, "%63s %lf", q, &period) == 2 && std::strcmp(q, "max") != 0) apply(std::atof(q), period);
        std::fclose(f);
    } else {
        double quota = -1, period = 0;
        if (FILE* a = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
            if (std::fscanf(a, "%lf", &quota) != 1) quota = -1;
            std::fclose(a);
        }
        if (FILE* b = std::fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) {
            if (std::fscanf(b, "%lf", &period) != 1) period = 0;
            std::fclose(b);
        }
        apply(quota, period);
    }
    // one process per GPU on a node (torch.distributed.run / torchrun export LOCAL_WORLD_SIZE): the quota -- or the
    // machine -- is shared by that many engines
    if (const char* lw = std::getenv("LOCAL_WORLD_SIZE")) {
        const int n = std::atoi(lw);
        if (n > 1) cores = std::max(1, cores / n);
    }
    return cores;
}
} // namespace

SelfplayStats SelfplayPipe::Run(double seconds) {
    // Hundreds of game threads build and drop a search tree per move.  Keep freed heap inside the process instead of
    // trimming / unmapping it: every munmap or brk shrink takes the address-space lock exclusively and stalls the page
    // faults of all other game threads.
    mallopt(M_TRIM_THRESHOLD, 1 << 30);
    mallopt(M_MMAP_THRESHOLD, 32 << 20);
    const auto t0 = std::chrono::steady_clock::now();
    writer_running_.store(true);
    std::thread writer([this]() { WriterLoop(); });
    std::vector<std::thread> workers;
    std::atomic<std::uint64_t> started{0};
    const int games = engine_.GetParallelGames();
    // The CPUs this process may use; game thread g is pinned to the g-th of them (round robin).  Unpinned, the 256
    // threads the pump wakes when a batch completes are queued next to the pump's CPU (wake-affine placement) and run
    // there one after another, which multiplies the time until the next batch is full.
    std::vector<int> cpus;
    {
        cpu_set_t set;
        CPU_ZERO(&set);
        if (sched_getaffinity(0, sizeof(set), &set) == 0)
            for (int c = 0; c < CPU_SETSIZE; ++c)
                if (CPU_ISSET(c, &set)) cpus.push_back(c);
    }
    for (int g = 0; g < games; ++g) engine_.search(g).SetAbortFlag(&stop_);
    // How the games are run: from 256 concurrent games on as fibers on a few threads per usable core (csrc/host/fiber.h):
    // a game blocks in exactly one place, the forward pipe, and that call yields to the thread's next game when it comes
    // from a fiber.  Fewer games: one OS thread per game (the reference's scheme, pipe.cc:235-296).  Measured on one
    // MI355X host (16 cores of CPU quota), 20b x 256, 400 visits: fibers 66.1 / 67.2 / 63.4 / 64.7 / 63.7 k evals/s at
    // 512 / 1024 / 2048 / 4096 / 8192 games, threads 57.7 / 59.5 / 18.5 / 7.4 k at 512 / 1024 / 2048 / 4096
    // (profiles/r02_selfplay_{fibers,threads}_g*.json).  `game_threads`: 0 = choose, N = N worker threads with the games
    // as fibers on them, -1 = always one thread per game.
    // Worker threads of the fiber mode: a few per core this process may actually use -- the CPUs of its affinity mask, cut
    // down to its cgroup CPU quota (the MI355X boxes of this pool show 256 CPUs and grant 16 cores of time: 256 polling
    // scheduler threads eat that quota in futex calls, 64 leave it to the games).
    const int cores = UsableCores(static_cast<int>(cpus.size()));
    int fiber_threads = 0;
    if (opt_.selfplay.game_threads > 0) fiber_threads = std::min(opt_.selfplay.game_threads, games);
    // ... and never more than 64: measured on the 16-CPU quota 64 threads 73.2 k evals/s at 4.7 cores, 96: 72.8 k at 5.0, 128: 72.0 k at
    // 5.8 (round 5) -- more scheduler threads only add wake-ups.  A box that grants more CPUs (an 8-GPU pod running ONE rank sees the
    // whole node's quota) must not get 512 threads for 512 games.
    else if (opt_.selfplay.game_threads == 0 && games >= 256) fiber_threads = std::min(games, std::min(64, std::max(8, 4 * cores)));
    // One process per GPU on a node: every rank would otherwise pin its threads to the same first CPUs of the (shared)
    // affinity mask.  Each local rank takes its own contiguous slice of the mask.
    size_t cpu_lo = 0, cpu_n = cpus.size();
    {
        const char* lw = std::getenv("LOCAL_WORLD_SIZE");
        const char* lr = std::getenv("LOCAL_RANK");
        const int n = lw ? std::atoi(lw) : 1, r = lr ? std::atoi(lr) : 0;
        if (n > 1 && r >= 0 && r < n && cpus.size() >= static_cast<size_t>(n)) {
            cpu_n = cpus.size() / static_cast<size_t>(n);
            cpu_lo = static_cast<size_t>(r) * cpu_n;
        }
    }
    const bool pin = std::getenv("SAYURI_NO_PIN") == nullptr;  // measuring aid: leave the worker threads to the scheduler
    auto thread_start = [&cpus, cpu_lo, cpu_n, pin](int idx) {
        if (pin && cpus.size() > 1) {
            cpu_set_t one;
            CPU_ZERO(&one);
            CPU_SET(cpus[cpu_lo + static_cast<size_t>(idx) % cpu_n], &one);
            pthread_setaffinity_np(pthread_self(), sizeof(one), &one);
        }
        // (no malloc warm-up: the search trees live in per-game TreeArenas that map their own slabs -- tree_arena.h -- and
        // what is left on the heap per game is small)
    };
    auto game_loop = [this, games, &started](int g) {
        try {
            bool first_game = true, halting = false;
            while (!stop_.load(std::memory_order_relaxed) && accumulation_games_.fetch_add(1) < max_games_.load(std::memory_order_relaxed)) {
                if (g == 0 && !halting) {  // pipe.cc:246-258: only the main worker looks, once per game
                    if (halt_wish_.load(std::memory_order_relaxed) || engine_.ShouldHalt()) {
                        halt_wish_.store(true, std::memory_order_relaxed);
                        if (!stats_cb_) WindDown();  // alone: act at once; with an exchange hook the verdict comes from there
                        halting = !stats_cb_;
                    }
                }
                started.fetch_add(1);
                auto item = std::make_shared<DataSgf>();
                engine_.PrepareGame(g);
                if (first_game && opt_.selfplay.stagger_moves > 0) {
                    const int before = engine_.state(g).GetMoveNumber();
                    engine_.PlayPolicyMoves(g, static_cast<int>(static_cast<long long>(g) * opt_.selfplay.stagger_moves / std::max(games, 1)));
                    prerolled_moves_.fetch_add(static_cast<std::uint64_t>(engine_.state(g).GetMoveNumber() - before));
                }
                first_game = false;
                // the game loop, abandoned between moves when the clock has run out
                Search& search = engine_.search(g);
                bool abandoned = false;
                while (engine_.Step(g)) {
                    if (stop_.load(std::memory_order_relaxed)) {
                        abandoned = true;
                        break;
                    }
                }
                if (abandoned) {
                    search.ClearTrainingBuffer();
                    break;
                }
                search.UpdateTerritoryHelper();
                engine_.GatherTrainingData(item->first, g);
                item->second = engine_.GatherSgfString(g);
                // a game is counted and handed to the writer in one step under the lock the window's end is taken under (Run):
                // what finishes after the window has closed is dropped like a game in progress, so that the games counted, the
                // moves counted for them and the chunks on disk are the same set
                std::lock_guard<std::mutex> lock(data_mu_);
                if (window_closed_) break;
                finished_moves_.fetch_add(static_cast<std::uint64_t>(engine_.state(g).GetMoveNumber()));
                const int played = played_games_.fetch_add(1) + 1;
                data_queue_.push_back(item);
                queries_queue_.emplace_back(played, std::string("hip ") + std::to_string(engine_.network().GetNumQueries()));
            }
        } catch (const std::exception& e) {
            // a failing backend ends the run: remember the first message, stop the other workers
            std::lock_guard<std::mutex> lock(data_mu_);
            if (error_.empty()) error_ = e.what();
            stop_.store(true);
        } catch (...) {  // nothing may leave a fiber's entry silently: the main loop would wait for its game forever
            std::lock_guard<std::mutex> lock(data_mu_);
            if (error_.empty()) error_ = "self-play worker: unknown exception";
            stop_.store(true);
        }
    };
    sayuri_fiber::FiberPool pool;
    if (fiber_threads > 0) {
        for (int g = 0; g < games; ++g) pool.Add([&game_loop, g] { game_loop(g); });
        workers.emplace_back([this, &pool, fiber_threads, &thread_start] {
            try {
                pool.Run(fiber_threads, thread_start);
            } catch (const std::exception& e) {  // e.g. a fiber stack that could not be mapped: end the run with an error
                std::lock_guard<std::mutex> lock(data_mu_);
                if (error_.empty()) error_ = std::string("fiber pool: ") + e.what();
                stop_.store(true);
            }
        });
    } else {
        for (int g = 0; g < games; ++g)
            workers.emplace_back([g, &thread_start, &game_loop] {
                thread_start(g);
                game_loop(g);
            });
    }
    auto snapshot = [&](SelfplayStats& st) {
        st.games_started = started.load();
        st.games_done = static_cast<std::uint64_t>(played_games_.load());
        st.moves = engine_.moves_played();
        st.playouts = engine_.playouts();
        st.nn_queries = engine_.network().GetNumQueries();
        st.cache_lookups = engine_.network().cache().lookups();
        st.cache_hits = engine_.network().cache().hits();
        st.finished_moves = finished_moves_.load();
        st.prerolled_moves = prerolled_moves_.load();
        st.chunks_saved_window = chunks_.load();
        st.writer_cpu_ns_window = writer_cpu_ns_.load();
        st.elapsed = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    };
    SelfplayStats st;
    bool timed_out = false;
    {
        // wake up often enough to stop close to the deadline; hand the counters to the exchange hook on its period
        double next_cb = stats_interval_;
        for (;;) {
            const double now = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (seconds > 0 && now >= seconds) break;
            if (played_games_.load() >= max_games_.load()) break;
            if (stop_.load()) break;  // a worker failed (error_ is set): report now, not at the deadline
            if (seconds <= 0 && accumulation_games_.load() >= max_games_.load() + games) break;  // every worker has left its loop
            if (stats_cb_ && now >= next_cb) {
                SelfplayStats snap;
                snapshot(snap);
                if (stats_cb_(&snap, halt_wish_.load() ? 1 : 0, stats_user_) != 0) WindDown();
                next_cb += stats_interval_;
            }
            std::this_thread::sleep_for(std::chrono::milliseconds(20));
        }
        if (seconds > 0 && played_games_.load() < max_games_.load() && !stop_.load()) {
            std::lock_guard<std::mutex> lock(data_mu_);
            window_closed_ = true;
            snapshot(st); // the window ends here: what the workers do while winding down is not counted
            timed_out = true;
        }
        if (seconds > 0 || stop_.load()) stop_.store(true);
    }
    for (auto& w : workers) w.join();
    if (std::getenv("SAYURI_MEMSTAT")) {  // where the resident set of a long run is (stderr, once, after the workers have ended)
        std::size_t slabs = 0, live = 0, biggest = 0;
        for (int g = 0; g < games; ++g) {
            const std::size_t b = engine_.search(g).arena_bytes();
            slabs += b;
            biggest = std::max(biggest, b);
            live += engine_.search(g).arena_live_blocks();
        }
        const struct mallinfo2 mi = mallinfo2();
        std::fprintf(stderr, "[memstat] games %d: tree arenas %.1f MB (largest %.1f MB, %zu live blocks); malloc: in use %.1f MB, free in arenas %.1f MB, mmapped %.1f MB\n",
                     games, slabs / 1048576.0, biggest / 1048576.0, live, mi.uordblks / 1048576.0, mi.fordblks / 1048576.0, mi.hblkhd / 1048576.0);
    }
    if (!timed_out) snapshot(st);
    const auto flush0 = std::chrono::steady_clock::now();
    writer_running_.store(false);
    writer.join();
    st.flush_ns = static_cast<std::uint64_t>(std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - flush0).count());
    st.writer_cpu_ns = writer_cpu_ns_.load() + flush_helper_cpu_ns_;
    st.bytes_written = bytes_written_.load();
    st.text_bytes = text_bytes_.load();
    if (!error_.empty()) throw std::runtime_error("self-play worker failed: " + error_);
    st.records = records_.load();
    st.chunks_saved = chunks_.load();
    return st;
}

} // namespace sayuri_engine
