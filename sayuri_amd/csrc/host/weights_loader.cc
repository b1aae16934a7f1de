// weights_loader.cc -- reads the reference's network-file format bit-exactly.
//
// Format (reference src/neural/loader.cc:67-121, written by train/torch/network.py:1399-1439):
//   get main / get info .. end / get stack .. end / get struct .. end / get parameters
//   <one tensor per "line"> / end parameters / end main
// A tensor is a text line of numbers, or with "FloatType float32bin" a little-endian f32
// stream terminated by the word 0xFFFFFFFF (loader.cc:833-898).  Layers arrive in the fixed
// order of DNNLoader::FillWeights / FillBlock (loader.cc:358-773); every convolution that is
// followed by a BatchNorm entry is folded here the way ProcessWeights does
// (loader.cc:775-793): b = (b - mean) * s, W *= s, s = 1/stddev (v1 files: 1/sqrt(var+1e-5),
// description.h:44-54).
#include "weights_model.h"

#include <algorithm>
#include <cctype>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <stdexcept>

namespace sayuri_host {

Activation StringToAct(std::string v) {
    for (char& c : v) c = static_cast<char>(std::tolower(static_cast<unsigned char>(c)));
    static const std::map<std::string, Activation> table = {
        {"identity", Activation::kIdentity}, {"relu", Activation::kReLU},   {"elu", Activation::kELU},
        {"selu", Activation::kSELU},         {"gelu", Activation::kGELU},   {"mish", Activation::kMISH},
        {"swish", Activation::kSwish},       {"hardswish", Activation::kHardSwish}};
    const auto it = table.find(v);
    if (it == table.end()) throw std::runtime_error{"Unknown activation type."};
    return it->second;
}

namespace {

struct LayerShape {
    enum Kind { kConv, kDepthwise, kBatchNorm, kFullyConnect } kind;
    std::vector<int> dims;
};

std::vector<std::string> Words(const std::string& line) {
    std::vector<std::string> out;
    std::istringstream ss(line);
    std::string w;
    while (ss >> w) out.push_back(w);
    return out;
}

class FileReader {
public:
    explicit FileReader(std::string data) : data_(std::move(data)) {}

    bool Line(std::string* out) {
        if (pos_ >= data_.size()) return false;
        const size_t nl = data_.find('\n', pos_);
        const size_t end = nl == std::string::npos ? data_.size() : nl;
        out->assign(data_, pos_, end - pos_);
        pos_ = nl == std::string::npos ? data_.size() : nl + 1;
        return true;
    }

    // one tensor from the parameter stream
    std::vector<float> Tensor(bool binary) {
        std::vector<float> v;
        if (binary) {
            for (;;) {
                if (pos_ + 4 > data_.size()) throw std::runtime_error{"unexpected end of the parameter stream"};
                std::uint32_t bits;
                std::memcpy(&bits, data_.data() + pos_, 4);  // file is little-endian, so is the host
                pos_ += 4;
                if (bits == 0xffffffffu) break;
                float f;
                std::memcpy(&f, &bits, 4);
                v.push_back(f);
            }
        } else {
            std::string line;
            if (!Line(&line)) throw std::runtime_error{"unexpected end of the parameter stream"};
            const char* p = line.c_str();
            for (;;) {
                char* end = nullptr;
                const double d = std::strtod(p, &end);  // parse as double, then narrow (loader.cc:851-854)
                if (end == p) break;
                v.push_back(static_cast<float>(d));
                p = end;
            }
        }
        return v;
    }

private:
    std::string data_;
    size_t pos_{0};
};

class Parser {
public:
    Parser(FileReader* rd, DNNWeights* w) : rd_(*rd), w_(*w) {}

    void Run() {
        std::string line;
        if (!rd_.Line(&line)) throw std::runtime_error{"weights file is empty"};
        {
            const auto ws = Words(line);
            if (ws.size() < 2 || ws[0] != "get" || ws[1] != "main")
                throw std::runtime_error{"weights file format is not acceptable"};
        }
        while (rd_.Line(&line)) {
            const auto ws = Words(line);
            if (ws.size() < 2 || ws[0] != "get") continue;
            if (ws[1] == "info") ReadInfo();
            else if (ws[1] == "stack") ReadStack();
            else if (ws[1] == "struct") ReadStruct();
            else if (ws[1] == "parameters") break;
        }
        Configure();
        ReadLayers();
        if (!rd_.Line(&line) || Words(line).empty() || Words(line)[0] != "end")
            throw std::runtime_error{"weights file format is not acceptable"};
        w_.loaded = true;
    }

private:
    // scope helpers: lines until "end", '#' comments skipped (loader.cc:123-147)
    template <typename F> void Scope(F&& f) {
        std::string line;
        while (rd_.Line(&line)) {
            const auto ws = Words(line);
            if (ws.empty() || ws[0][0] == '#') continue;
            if (ws[0] == "end") return;
            f(ws);
        }
    }
    void ReadInfo() {
        Scope([&](const std::vector<std::string>& ws) {
            if (ws.size() >= 2) info_.emplace(ws[0], ws[1]);
        });
    }
    void ReadStack() {
        Scope([&](const std::vector<std::string>& ws) { stack_.push_back(ws[0]); });
    }
    void ReadStruct() {  // loader.cc:149-188
        Scope([&](const std::vector<std::string>& ws) {
            LayerShape s;
            for (size_t i = 1; i < ws.size(); ++i) s.dims.push_back(std::stoi(ws[i]));
            if (ws[0] == "FullyConnect" && s.dims.size() == 2) s.kind = LayerShape::kFullyConnect;
            else if (ws[0] == "Convolution" && s.dims.size() == 3) s.kind = LayerShape::kConv;
            else if (ws[0] == "DepthwiseConvolution" && s.dims.size() == 3) s.kind = LayerShape::kDepthwise;
            else if (ws[0] == "BatchNorm" && s.dims.size() == 1) s.kind = LayerShape::kBatchNorm;
            else throw std::runtime_error{"layer shape is error"};
            shapes_.push_back(s);
        });
    }

    bool Has(const char* key) const { return info_.count(key) != 0; }
    int InfoInt(const char* key) const {
        const auto it = info_.find(key);
        if (it == info_.end()) throw std::runtime_error{std::string("missing info key ") + key};
        return std::stoi(it->second);
    }

    void Configure() {  // loader.cc:190-316 + 628-643
        binary_ = Has("FloatType") && info_.at("FloatType") == "float32bin";
        w_.version = Has("Version") ? InfoInt("Version") : 1;
        if (w_.version >= 6) throw std::runtime_error{"do not support this version"};
        if (w_.version >= 3) {
            w_.input_channels = 43; w_.probabilities_channels = 5; w_.pass_probability_outputs = 5;
            w_.ownership_channels = 1; w_.value_misc_outputs = 15;
        } else {
            w_.input_channels = 38; w_.probabilities_channels = 1; w_.pass_probability_outputs = 1;
            w_.ownership_channels = 1; w_.value_misc_outputs = 5;
        }
        w_.policy_head_type = PolicyHeadType::kNormal;
        if (Has("PolicyHeadType")) {
            std::string t = info_.at("PolicyHeadType");
            for (char& c : t) c = static_cast<char>(std::tolower(static_cast<unsigned char>(c)));
            if (t == "replk") w_.policy_head_type = PolicyHeadType::kRepLK;
            else if (t != "normal") throw std::runtime_error{"unknown policy head type"};
        }
        w_.default_act = Has("ActivationFunction") ? StringToAct(info_.at("ActivationFunction")) : Activation::kReLU;
        w_.residual_blocks = InfoInt("ResidualBlocks");
        w_.residual_channels = InfoInt("ResidualChannels");
        w_.policy_head_channels = InfoInt(w_.version >= 5 ? "PolicyHeadChannels" : "PolicyExtract");
        w_.value_head_channels = InfoInt(w_.version >= 5 ? "ValueHeadChannels" : "ValueExtract");
        if (w_.input_channels != InfoInt("InputChannels"))
            throw std::runtime_error{"the number of input channels is wrong"};

        if (stack_.empty()) {  // legacy files: ResidualBlock[-SE] only (loader.cc:267-292)
            size_t inner = 0;
            for (int b = 0; b < w_.residual_blocks; ++b) {
                inner += 4;
                std::string name = "ResidualBlock";
                if (inner + 2 < shapes_.size() && shapes_[inner + 2].kind == LayerShape::kFullyConnect) {
                    name += "-SE";
                    inner += 2;
                }
                stack_.push_back(name);
            }
            if (shapes_.size() != 10 + inner + 2) throw std::runtime_error{"do not support this weights format"};
        }
        if (static_cast<int>(stack_.size()) < w_.residual_blocks)
            throw std::runtime_error{"the stack is shorter than ResidualBlocks"};
    }

    const LayerShape& NextShape(LayerShape::Kind a, LayerShape::Kind b) {
        if (cursor_ >= shapes_.size()) throw std::runtime_error{"the struct list is too short"};
        const LayerShape& s = shapes_[cursor_++];
        if (s.kind != a && s.kind != b) throw std::runtime_error{"unexpected layer kind in struct"};
        return s;
    }

    // conv (+ optional BatchNorm folded in place)
    void ReadConv(ConvLayer* conv, bool with_bn) {
        const LayerShape& s = NextShape(LayerShape::kConv, LayerShape::kDepthwise);
        conv->Set(s.dims[0], s.dims[1], s.dims[2]);
        auto& W = conv->GetWeights();
        auto& B = conv->GetBiases();
        W = rd_.Tensor(binary_);
        B = rd_.Tensor(binary_);
        const size_t expect = static_cast<size_t>(s.dims[0]) * s.dims[1] * s.dims[2] * s.dims[2];
        if (W.size() != expect) throw std::runtime_error{"the weights size of convolutional layer is not acceptable"};
        if (B.size() != static_cast<size_t>(s.dims[1]))
            throw std::runtime_error{"the biases size of convolutional layer is not acceptable"};
        if (!with_bn) return;
        const LayerShape& bn = NextShape(LayerShape::kBatchNorm, LayerShape::kBatchNorm);
        const auto means = rd_.Tensor(binary_);
        const auto spread = rd_.Tensor(binary_);
        const size_t K = B.size();
        if (bn.dims[0] != s.dims[1] || means.size() != K || spread.size() != K)
            throw std::runtime_error{"the size of batch normalization layer is not acceptable"};
        const size_t per_out = W.size() / K;
        for (size_t o = 0; o < K; ++o) {
            const float inv = w_.version == 1 ? 1.0f / std::sqrt(spread[o] + 1e-5f) : 1.0f / spread[o];
            B[o] -= means[o];
            for (size_t i = 0; i < per_out; ++i) W[o * per_out + i] *= inv;
            B[o] *= inv;
        }
    }

    void ReadFc(LinearLayer* fc) {
        const LayerShape& s = NextShape(LayerShape::kFullyConnect, LayerShape::kFullyConnect);
        fc->Set(s.dims[0], s.dims[1]);
        fc->GetWeights() = rd_.Tensor(binary_);
        fc->GetBiases() = rd_.Tensor(binary_);
        if (fc->GetWeights().size() != static_cast<size_t>(s.dims[0]) * s.dims[1])
            throw std::runtime_error{"the weights size of linear layer is not acceptable"};
        if (fc->GetBiases().size() != static_cast<size_t>(s.dims[1]))
            throw std::runtime_error{"the biases size of linear layer is not acceptable"};
    }

    static void Require(bool ok, const char* what) {
        if (!ok) throw std::runtime_error{what};
    }

    void ReadBlock(const std::string& stack_name) {  // loader.cc:358-626
        auto blk = std::make_unique<BlockBasic>();
        std::string name = stack_name;
        std::replace(name.begin(), name.end(), '-', ' ');
        for (const auto& part : Words(name)) {
            if (part == "ResidualBlock") blk->type = BlockBasic::kResidualBlock;
            else if (part == "BottleneckBlock") blk->type = BlockBasic::kBottleneckBlock;
            else if (part == "NestedBottleneckBlock") blk->type = BlockBasic::kNestedBottleneckBlock;
            else if (part == "MixerBlock") blk->type = BlockBasic::kMixerBlock;
            else if (part == "SE") blk->apply_se = true;
            else if (part == "FixUp") {}
            else throw std::runtime_error{"do not support this block type [" + stack_name + "]"};
        }
        const int C = w_.residual_channels;
        auto is = [](const ConvLayer& c, int in, int out, int k) {
            return c.GetInputs() == in && c.GetOutputs() == out && c.GetFilter() == k;
        };
        switch (blk->type) {
        case BlockBasic::kResidualBlock:
            ReadConv(&blk->conv1, true);
            ReadConv(&blk->conv2, true);
            Require(is(blk->conv1, C, C, 3) && is(blk->conv2, C, C, 3), "the residual block is wrong");
            break;
        case BlockBasic::kBottleneckBlock:
        case BlockBasic::kNestedBottleneckBlock: {
            ReadConv(&blk->pre_btl_conv, true);
            ReadConv(&blk->conv1, true);
            ReadConv(&blk->conv2, true);
            if (blk->type == BlockBasic::kNestedBottleneckBlock) {
                ReadConv(&blk->conv3, true);
                ReadConv(&blk->conv4, true);
            }
            ReadConv(&blk->post_btl_conv, true);
            const int I = blk->pre_btl_conv.GetOutputs();
            blk->bottleneck_channels = I;
            Require(is(blk->pre_btl_conv, C, I, 1) && is(blk->post_btl_conv, I, C, 1),
                    "the outer channels of bottleneck block is wrong");
            Require(is(blk->conv1, I, I, 3) && is(blk->conv2, I, I, 3), "the inner channels of bottleneck block is wrong");
            if (blk->type == BlockBasic::kNestedBottleneckBlock)
                Require(is(blk->conv3, I, I, 3) && is(blk->conv4, I, I, 3),
                        "the inner channels of nested bottleneck block is wrong");
            break;
        }
        case BlockBasic::kMixerBlock: {
            ReadConv(&blk->dw_conv, true);
            ReadConv(&blk->conv1, true);
            ReadConv(&blk->conv2, true);
            const int F = blk->conv1.GetOutputs();
            blk->feedforward_channels = F;
            Require(blk->dw_conv.GetOutputs() == C && is(blk->conv1, C, F, 1) && is(blk->conv2, F, C, 1),
                    "the channels of mixer block is wrong");
            break;
        }
        default:
            throw std::runtime_error{"need the ResidualBlock, BottleneckBlock, NestedBottleneckBlock or MixerBlock"};
        }
        if (blk->apply_se) {
            ReadFc(&blk->squeeze);
            ReadFc(&blk->excite);
            blk->se_size = blk->squeeze.GetOutputs();
            Require(blk->squeeze.GetInputs() == 3 * C && blk->excite.GetOutputs() == 2 * C, "the SE module size is wrong");
        }
        w_.tower.push_back(std::move(blk));
    }

    void ReadLayers() {  // loader.cc:658-761
        ReadConv(&w_.input_conv, true);
        Require(w_.input_conv.GetInputs() == w_.input_channels && w_.input_conv.GetOutputs() == w_.residual_channels &&
                    w_.input_conv.GetFilter() == 3,
                "the input layers are wrong");
        for (int b = 0; b < w_.residual_blocks; ++b) ReadBlock(stack_[b]);

        ReadConv(&w_.p_hd_conv, true);
        if (w_.policy_head_type == PolicyHeadType::kRepLK) {
            ReadConv(&w_.p_dw_conv, true);
            ReadConv(&w_.p_pt_conv, true);
        }
        ReadFc(&w_.p_inter_fc);
        ReadConv(&w_.prob_conv, false);
        ReadFc(&w_.pass_fc);
        Require(w_.p_hd_conv.GetFilter() == 1 && w_.prob_conv.GetFilter() == 1,
                "the policy convolution kernel size is wrong");
        Require(w_.prob_conv.GetOutputs() == w_.probabilities_channels, "the number of policy ouput size is wrong");
        Require(w_.p_inter_fc.GetOutputs() == w_.pass_fc.GetInputs() &&
                    w_.p_inter_fc.GetInputs() == 3 * w_.policy_head_channels &&
                    w_.p_inter_fc.GetOutputs() == w_.policy_head_channels,
                "the number of policy fully connect size is wrong");
        Require(w_.pass_fc.GetOutputs() == w_.pass_probability_outputs, "the number of pass ouput size is wrong");

        ReadConv(&w_.v_hd_conv, true);
        ReadFc(&w_.v_inter_fc);
        ReadConv(&w_.v_ownership, false);
        ReadFc(&w_.v_misc);
        Require(w_.v_hd_conv.GetFilter() == 1 && w_.v_ownership.GetFilter() == 1,
                "the value convolution kernel size is wrong");
        Require(w_.v_ownership.GetOutputs() == w_.ownership_channels, "the number of ownership ouput size is wrong");
        Require(w_.v_inter_fc.GetOutputs() == w_.v_misc.GetInputs() &&
                    w_.v_inter_fc.GetInputs() == 3 * w_.value_head_channels &&
                    w_.v_inter_fc.GetOutputs() == 3 * w_.value_head_channels,
                "the number of value fully connect size is wrong");
        Require(w_.v_misc.GetOutputs() == w_.value_misc_outputs, "the misc value layer size is wrong.");
    }

    FileReader& rd_;
    DNNWeights& w_;
    std::map<std::string, std::string> info_;
    std::vector<std::string> stack_;
    std::vector<LayerShape> shapes_;
    size_t cursor_{0};
    bool binary_{false};
};

}  // namespace

bool LoadWeightsFile(const std::string& filename, DNNWeights* weights, std::string* error) {
    weights->loaded = false;
    try {
        if (filename.empty()) throw std::runtime_error{"There is no weights file."};
        std::ifstream file(filename, std::ifstream::binary | std::ifstream::in);
        if (!file.is_open()) throw std::runtime_error{"Couldn't open weights file from " + filename + "!"};
        std::stringstream buffer;
        buffer << file.rdbuf();
        FileReader rd(buffer.str());
        Parser(&rd, weights).Run();
        const auto slash = filename.find_last_of("/\\");
        weights->name = slash == std::string::npos ? filename : filename.substr(slash + 1);
        if (weights->name.empty()) weights->name = "network";
        return true;
    } catch (const std::exception& e) {
        if (error) *error = e.what();
        weights->loaded = false;
        return false;
    }
}

}  // namespace sayuri_host
