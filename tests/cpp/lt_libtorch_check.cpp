// Loads archives written by rlgpu_lt_write_model / rlgpu_lt_write_adam with the real libtorch, the way the reference does
// (TorchLoadSaveSeq: torch::load(seq, ifstream, device), PPOLearner.cpp:372-386; optimizer: InputArchive::load_from + Adam::load,
// :453-458), and dumps what it got as raw fp32 for the test to compare: [params | exp_avg | exp_avg_sq | step | lr].
#include <torch/torch.h>
#include <fstream>
int main(int argc, char** argv) {
    if (argc < 4) return 2;
    torch::nn::Sequential seq;
    seq->push_back(torch::nn::Linear(7, 6)); seq->push_back(torch::nn::ReLU());
    seq->push_back(torch::nn::Linear(6, 5)); seq->push_back(torch::nn::ReLU());
    seq->push_back(torch::nn::Linear(5, 3));
    {
        std::ifstream in(argv[1], std::ios::binary);
        in >> std::noskipws;
        torch::load(seq, in, torch::kCPU);
    }
    torch::optim::Adam opt(seq->parameters(), torch::optim::AdamOptions(1.0));
    torch::serialize::InputArchive ar;
    ar.load_from(std::string(argv[2]), torch::kCPU);
    opt.load(ar);
    std::ofstream e(argv[3], std::ios::binary);
    auto dump = [&](const torch::Tensor& t) { auto c = t.detach().contiguous().to(torch::kFloat32); e.write((const char*)c.data_ptr<float>(), c.numel() * 4); };
    for (auto& p : seq->parameters()) dump(p);
    float step = 0;
    for (int k = 0; k < 2; k++)
        for (auto& p : seq->parameters()) {
            auto& st = static_cast<torch::optim::AdamParamState&>(*opt.state().at(p.unsafeGetTensorImpl()));
            dump(k ? st.exp_avg_sq() : st.exp_avg());
            step = (float)st.step();
        }
    e.write((const char*)&step, 4);
    float lr = (float)static_cast<torch::optim::AdamOptions&>(opt.param_groups()[0].options()).lr();
    e.write((const char*)&lr, 4);
    return 0;
}
