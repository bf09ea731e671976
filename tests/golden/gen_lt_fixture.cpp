// Generates tests/golden/lt_{model,optim}.lt + lt_expected.f32 with the real libtorch (the pip wheel's C++ API), the same calls the
// reference makes: torch::save(nn::Sequential, ofstream) (PPOLearner.cpp:408-411) and Adam::save(OutputArchive) + save_to
// (PPOLearner.cpp:466-472).  Build + run: see tests/golden/gen_lt_fixture.sh.  Layout of lt_expected.f32 (little-endian fp32):
// [params in state-dict order | exp_avg | exp_avg_sq | step as one float].
#include <torch/torch.h>
#include <fstream>
int main(int argc, char** argv) {
    std::string dir = argc > 1 ? argv[1] : ".";
    torch::manual_seed(7);
    torch::nn::Sequential seq;   // DiscretePolicy.cpp:10-26 with layer sizes {6, 5}, 7 inputs, 3 actions
    seq->push_back(torch::nn::Linear(7, 6)); seq->push_back(torch::nn::ReLU());
    seq->push_back(torch::nn::Linear(6, 5)); seq->push_back(torch::nn::ReLU());
    seq->push_back(torch::nn::Linear(5, 3));
    torch::optim::Adam opt(seq->parameters(), torch::optim::AdamOptions(2e-4));
    for (int i = 0; i < 4; i++) {
        opt.zero_grad();
        auto y = seq->forward(torch::randn({16, 7})).pow(2).sum();
        y.backward();
        opt.step();
    }
    { std::ofstream o(dir + "/lt_model.lt", std::ios::binary); torch::save(seq, o); }
    torch::serialize::OutputArchive a; opt.save(a); a.save_to(dir + "/lt_optim.lt");
    std::ofstream e(dir + "/lt_expected.f32", std::ios::binary);
    auto dump = [&](const torch::Tensor& t) { auto c = t.detach().contiguous().to(torch::kFloat32); e.write((const char*)c.data_ptr<float>(), c.numel() * 4); };
    for (auto& p : seq->parameters()) dump(p);
    for (int k = 0; k < 2; k++)
        for (auto& p : seq->parameters()) {
            auto& st = static_cast<torch::optim::AdamParamState&>(*opt.state().at(p.unsafeGetTensorImpl()));
            dump(k ? st.exp_avg_sq() : st.exp_avg());
        }
    float step = 4; e.write((const char*)&step, 4);
}
