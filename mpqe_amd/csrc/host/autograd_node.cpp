// _autograd_node -- the autograd node of a drop-in margin_loss call (mpqe_amd/dropin.py) as a C++ torch::autograd::Node.
//
// The reference's training step builds `loss = l_0 + w_1 l_1 + ... + w_10 l_10` from eleven margin_loss results and calls
// loss.backward() (train_helpers.py:81-119). Behind the drop-in every l_i is a forward-only library call whose backward is
// deferred: the node only remembers (call number, upstream gradient); when the engine has walked the graph, ONE callback runs
// all of them as one fused step. As a Python torch.autograd.Function each node costs ~10 us of interpreter time twice per
// step (apply at the call, the engine's trip into Python at backward); here the engine stays in C++ until the pass' single
// callback takes the GIL once.
//
//   p = Pass(flush)                        flush(ids: list[int], grads: list[Tensor]) -- called once per backward pass
//   out = make_loss(p, call_id, buf)       0-dim tensor over buf's first element (its own tensor, not an autograd view: the
//                                          reference adds into the first loss in place), grad_fn = the node
//   p.take_dead() -> list[int]             call ids whose nodes have been destroyed (their graphs freed) since the last call
//   storage_use_count(t) -> int            holders of t's storage
//
// No arithmetic of the data path lives here.
#include <torch/extension.h>
#include <torch/csrc/autograd/engine.h>
#include <torch/csrc/autograd/function.h>
#include <torch/csrc/autograd/variable.h>

#include <mutex>
#include <vector>

namespace {

struct Pass {
    std::mutex mu;
    std::vector<int64_t> ids;
    std::vector<at::Tensor> grads;
    std::vector<int64_t> dead;
    bool scheduled = false;
    py::object flush;
    explicit Pass(py::object f) : flush(std::move(f)) {}
    ~Pass() {
        py::gil_scoped_acquire gil;
        flush = py::object();
    }
    std::vector<int64_t> take_dead() {
        std::lock_guard<std::mutex> lock(mu);
        std::vector<int64_t> out;
        out.swap(dead);
        return out;
    }
};

struct MarginLossNode : public torch::autograd::Node {
    std::shared_ptr<Pass> pass;
    int64_t id = 0;

    torch::autograd::variable_list apply(torch::autograd::variable_list &&grads) override {
        bool schedule = false;
        {
            std::lock_guard<std::mutex> lock(pass->mu);
            pass->ids.push_back(id);
            pass->grads.push_back(grads.empty() ? at::Tensor() : grads[0]);
            if (!pass->scheduled) {
                pass->scheduled = true;
                schedule = true;
            }
        }
        if (schedule) {
            std::shared_ptr<Pass> p = pass;
            torch::autograd::Engine::get_default_engine().queue_callback([p]() {
                std::vector<int64_t> ids;
                std::vector<at::Tensor> g;
                {
                    std::lock_guard<std::mutex> lock(p->mu);
                    ids.swap(p->ids);
                    g.swap(p->grads);
                    p->scheduled = false;
                }
                py::gil_scoped_acquire gil;
                p->flush(ids, g);
            });
        }
        return {};
    }

    ~MarginLossNode() override {
        if (pass) {
            std::lock_guard<std::mutex> lock(pass->mu);
            pass->dead.push_back(id);
        }
    }
};

at::Tensor make_loss(const std::shared_ptr<Pass> &pass, int64_t id, const at::Tensor &buf) {
    at::Tensor out = at::empty({0}, buf.options());
    out.set_(buf.storage(), buf.storage_offset(), {}, {});
    auto node = std::shared_ptr<MarginLossNode>(new MarginLossNode(), torch::autograd::deleteNode);
    node->pass = pass;
    node->id = id;
    torch::autograd::create_gradient_edge(out, node);
    return out;
}

// holders of a tensor's storage (the drop-in's per-lane pool of loss words re-uses one only while it holds the last reference)
int64_t storage_use_count(const at::Tensor &t) { return static_cast<int64_t>(t.storage().use_count()); }

}  // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
    py::class_<Pass, std::shared_ptr<Pass>>(m, "Pass")
        .def(py::init<py::object>())
        .def("take_dead", &Pass::take_dead);
    m.def("storage_use_count", &storage_use_count, "number of holders of the tensor's storage");
    m.def("make_loss", &make_loss, "0-dim loss tensor over buf[0] whose grad_fn defers to the pass' flush");
}
