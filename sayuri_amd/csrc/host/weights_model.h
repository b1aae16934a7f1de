// weights_model.h -- host-side network description consumed by HipForwardPipe.
//
// Same member / accessor names as the reference's DNNWeights family
// (src/neural/description.h:11-215) so hip_forward_pipe.cc compiles unchanged against either
// this header (stand-alone library) or the reference's (in-tree drop-in).
// The tensors are the BN-folded fp32 arrays the reference loader leaves behind
// (loader.cc:775-831): conv weights [K][C][k][k], biases [K], fc weights [out][in].
// This backend never needs the Winograd-transformed copy, so none is kept.
#pragma once

#include <memory>
#include <string>
#include <vector>

namespace sayuri_host {

// reference src/neural/activation.h:8-17
enum class Activation : int {
    kIdentity = 0, kReLU = 1, kELU = 2, kSELU = 3, kGELU = 4, kMISH = 5, kSwish = 6, kHardSwish = 7
};
Activation StringToAct(std::string name);  // throws std::runtime_error on unknown names

class LinearLayer {
public:
    void Set(int inputs, int outputs) { inputs_ = inputs; outputs_ = outputs; }
    int GetInputs() const { return inputs_; }
    int GetOutputs() const { return outputs_; }
    std::vector<float>& GetWeights() { return weights_; }
    std::vector<float>& GetBiases() { return biases_; }
private:
    std::vector<float> weights_, biases_;
    int inputs_{0}, outputs_{0};
};

class ConvLayer {
public:
    void Set(int inputs, int outputs, int filter) { inputs_ = inputs; outputs_ = outputs; filter_ = filter; }
    int GetInputs() const { return inputs_; }
    int GetOutputs() const { return outputs_; }
    int GetFilter() const { return filter_; }
    std::vector<float>& GetWeights() { return weights_; }
    std::vector<float>& GetBiases() { return biases_; }
private:
    std::vector<float> weights_, biases_;
    int inputs_{0}, outputs_{0}, filter_{0};
};

class BlockBasic {
public:
    enum Type { kUnknown, kResidualBlock, kBottleneckBlock, kNestedBottleneckBlock, kMixerBlock };
    bool IsResidualBlock() const { return type == kResidualBlock; }
    bool IsBottleneckBlock() const { return type == kBottleneckBlock; }
    bool IsNestedBottleneckBlock() const { return type == kNestedBottleneckBlock; }
    bool IsMixerBlock() const { return type == kMixerBlock; }

    Type type{kUnknown};
    ConvLayer conv1, conv2, conv3, conv4;
    ConvLayer pre_btl_conv, post_btl_conv;
    int bottleneck_channels{0};
    ConvLayer dw_conv;
    int feedforward_channels{0};
    LinearLayer squeeze, excite;
    int se_size{0};
    bool apply_se{false};
};

enum class PolicyHeadType { kNormal, kRepLK };

class DNNWeights {
public:
    std::string name;
    int version{-1};
    bool loaded{false};
    bool winograd{false};  // kept for interface parity; unused by the MI355X backend

    int input_channels{0};
    int residual_blocks{0};
    int residual_channels{0};
    PolicyHeadType policy_head_type{PolicyHeadType::kNormal};
    int policy_head_channels{0};
    int probabilities_channels{0};
    int pass_probability_outputs{0};
    int value_head_channels{0};
    int ownership_channels{0};
    int value_misc_outputs{0};
    Activation default_act{Activation::kReLU};

    ConvLayer input_conv;
    std::vector<std::unique_ptr<BlockBasic>> tower;
    ConvLayer p_hd_conv;
    LinearLayer p_inter_fc;
    ConvLayer p_dw_conv, p_pt_conv;
    ConvLayer prob_conv;
    LinearLayer pass_fc;
    ConvLayer v_hd_conv;
    LinearLayer v_inter_fc;
    ConvLayer v_ownership;
    LinearLayer v_misc;
};

// Parse a Sayuri network file (text or float32bin) and fold batch-norm into the convolutions.
// Counterpart of DNNLoader::FromFile (loader.cc:26-65) with one deliberate difference in error
// behaviour: the reference logs and leaves weights->loaded == false; here the cause is also
// returned so callers can surface it.  Never throws.
bool LoadWeightsFile(const std::string& filename, DNNWeights* weights, std::string* error);

}  // namespace sayuri_host
