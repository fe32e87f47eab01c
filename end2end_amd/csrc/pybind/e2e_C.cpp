// end2end_amd._C -- the thin pybind11 layer between the Python host code and the C ABI of libe2e_ctc.so
// (include/e2e_ctc.h).  It stands where the reference's pybind modules stand (src/losses/ctc_loss_py.cpp:5-17,
// src/decoders/ctc_decoder_py.cpp:5-39) but carries no tensor types: the Python side hands device addresses
// (tensor.data_ptr()), strides and sizes; every function forwards to exactly one extern "C" entry point and turns a
// negative return code into the Python exception E2EError carrying e2e_last_error().
//
// Compiled with plain g++ against the pybind11 headers (no torch headers, no HIP headers): end2end_amd/csrc/Makefile.
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <cstdint>
#include <string>
#include <vector>

#include "../../../include/e2e_ctc.h"

namespace py = pybind11;

namespace {

struct E2EError : std::runtime_error {
  using std::runtime_error::runtime_error;
};

inline void check(int rc) {
  if (rc != E2E_OK) throw E2EError(std::string("libe2e_ctc: ") + e2e_last_error() + " (code " + std::to_string(rc) + ")");
}

template <typename T>
inline T* ptr(uintptr_t a) { return reinterpret_cast<T*>(a); }

// Owner of an e2e_lm handle (the decoder owns its KenLM model through a unique_ptr upstream, ctc_decoder.h:51).
class LanguageModel {
 public:
  LanguageModel(const std::string& path, const std::vector<std::string>& labels, bool case_sensitive) {
    std::vector<const char*> c;
    c.reserve(labels.size());
    for (const auto& s : labels) c.push_back(s.c_str());
    check(e2e_lm_load_arpa(path.c_str(), c.data(), (int)c.size(), case_sensitive ? 1 : 0, &lm_));
  }
  ~LanguageModel() { e2e_lm_free(lm_); }
  LanguageModel(const LanguageModel&) = delete;
  LanguageModel& operator=(const LanguageModel&) = delete;

  uintptr_t handle() const { return reinterpret_cast<uintptr_t>(lm_); }
  int order() const { return e2e_lm_order(lm_); }
  int device() const { return e2e_lm_device(lm_); }
  uint32_t word_index(const std::string& w) const { return e2e_lm_word_index(lm_, w.c_str()); }
  double score(const std::vector<uint32_t>& ctx, uint32_t word) const {
    return e2e_lm_score(lm_, ctx.data(), (int)ctx.size(), word);
  }

 private:
  e2e_lm* lm_ = nullptr;
};

}  // namespace

PYBIND11_MODULE(_C, m) {
  m.doc() = "pybind11 layer over the C ABI of libe2e_ctc.so (include/e2e_ctc.h): addresses and sizes in, exceptions out";
  py::register_exception<E2EError>(m, "E2EError", PyExc_RuntimeError);

  m.attr("ABI_VERSION") = E2E_CTC_ABI_VERSION;
  m.attr("F32") = E2E_F32;
  m.attr("F64") = E2E_F64;
  m.attr("F16") = E2E_F16;
  m.attr("BF16") = E2E_BF16;
  m.attr("ERR_UNSUPPORTED") = E2E_ERR_UNSUPPORTED;
  m.attr("ALGO_AUTO") = E2E_ALGO_AUTO;
  m.attr("ALGO_EXACT") = E2E_ALGO_EXACT;
  m.attr("ALGO_FAST") = E2E_ALGO_FAST;
  m.attr("REDUCE_NONE") = E2E_REDUCE_NONE;
  m.attr("REDUCE_SUM") = E2E_REDUCE_SUM;
  m.attr("REDUCE_MEAN") = E2E_REDUCE_MEAN;
  m.attr("CHAINS_F64") = E2E_CHAINS_F64;
  m.attr("CHAINS_F32") = E2E_CHAINS_F32;

  m.def("abi_version", [] { return e2e_ctc_abi_version(); });
  m.def("last_error", [] { return std::string(e2e_last_error()); });

  m.def("ctc_loss_workspace_bytes", [](int B, int T, int V, int Smax, int dtype, int algo) {
    return e2e_ctc_loss_workspace_bytes(B, T, V, Smax, dtype, algo);
  });

  m.def("ctc_loss_takes_dtype", [](int dtype, int algo, int T, int V, int Smax, int64_t sB, int64_t sT, int64_t sV, uintptr_t x, uintptr_t grads) {
    return e2e_ctc_loss_takes_dtype(dtype, algo, T, V, Smax, sB, sT, sV, reinterpret_cast<const void*>(x), reinterpret_cast<const void*>(grads)) != 0;
  });

  // (grad_scale / reduced / reduction / chains: e2e_ctc_loss_opts; the defaults are the plain call)
  m.def("ctc_loss_fwd_bwd",
        [](uintptr_t x, int dtype, bool input_is_logprobs, int64_t sB, int64_t sT, int64_t sV, uintptr_t targets,
           int64_t tgt_stride, uintptr_t x_len, uintptr_t t_len, int B, int T, int V, int Smax, int blank,
           uintptr_t losses, uintptr_t grads, uintptr_t workspace, size_t workspace_bytes, int algo, uintptr_t stream,
           double grad_scale, uintptr_t reduced, int reduction, int chains) {
          e2e_ctc_loss_opts o{grad_scale, ptr<void>(reduced), reduction, chains};
          check(e2e_ctc_loss_fwd_bwd_opt(ptr<const void>(x), dtype, input_is_logprobs ? 1 : 0, sB, sT, sV,
                                         ptr<const int64_t>(targets), tgt_stride, ptr<const int64_t>(x_len),
                                         ptr<const int64_t>(t_len), B, T, V, Smax, blank, ptr<void>(losses),
                                         ptr<void>(grads), ptr<void>(workspace), workspace_bytes, algo,
                                         ptr<void>(stream), &o));
        },
        py::arg("x"), py::arg("dtype"), py::arg("input_is_logprobs"), py::arg("sB"), py::arg("sT"), py::arg("sV"),
        py::arg("targets"), py::arg("tgt_stride"), py::arg("x_len"), py::arg("t_len"), py::arg("B"), py::arg("T"),
        py::arg("V"), py::arg("Smax"), py::arg("blank"), py::arg("losses"), py::arg("grads"), py::arg("workspace"),
        py::arg("workspace_bytes"), py::arg("algo"), py::arg("stream"), py::arg("grad_scale") = 1.0,
        py::arg("reduced") = 0, py::arg("reduction") = E2E_REDUCE_NONE, py::arg("chains") = E2E_CHAINS_F64);

  m.def("ctc_scale_grads",
        [](uintptr_t grads, int dtype, uintptr_t scale, int B, int64_t row_elems, uintptr_t stream) {
          check(e2e_ctc_scale_grads(ptr<void>(grads), dtype, ptr<const void>(scale), B, row_elems, ptr<void>(stream)));
        },
        py::arg("grads"), py::arg("dtype"), py::arg("scale"), py::arg("B"), py::arg("row_elems"), py::arg("stream"));

  m.def("ctc_greedy",
        [](uintptr_t x, int dtype, int64_t sB, int64_t sT, int64_t sV, uintptr_t x_len, int B, int T, int V, int blank,
           uintptr_t out, uintptr_t out_len, uintptr_t stream) {
          check(e2e_ctc_greedy(ptr<const void>(x), dtype, sB, sT, sV, ptr<const int64_t>(x_len), B, T, V, blank,
                               ptr<int64_t>(out), ptr<int64_t>(out_len), ptr<void>(stream)));
        },
        py::arg("x"), py::arg("dtype"), py::arg("sB"), py::arg("sT"), py::arg("sV"), py::arg("x_len"), py::arg("B"),
        py::arg("T"), py::arg("V"), py::arg("blank"), py::arg("out"), py::arg("out_len"), py::arg("stream"));

  m.def("ctc_beam_workspace_bytes",
        [](int B, int T, int V, int beam_width) { return e2e_ctc_beam_workspace_bytes(B, T, V, beam_width); });
  m.def("ctc_beam_workspace_bytes_lm", [](int B, int T, int V, int beam_width, bool with_lm) {
    return e2e_ctc_beam_workspace_bytes_lm(B, T, V, beam_width, with_lm ? 1 : 0);
  });

  m.def("ctc_beam_max_width", [](int V, bool with_lm) { return e2e_ctc_beam_max_width(V, with_lm ? 1 : 0); });

  m.def("ctc_beam",
        [](uintptr_t lp, int dtype, int64_t sB, int64_t sT, int64_t sV, uintptr_t x_len, int B, int T, int V, int blank,
           int beam_width, int space_id, uintptr_t lm, double lmwt, double wip, double oov_penalty, uintptr_t out,
           int64_t max_out, uintptr_t out_len, uintptr_t workspace, size_t workspace_bytes, uintptr_t stream) {
          check(e2e_ctc_beam(ptr<const void>(lp), dtype, sB, sT, sV, ptr<const int64_t>(x_len), B, T, V, blank,
                             beam_width, space_id, ptr<const e2e_lm>(lm), lmwt, wip, oov_penalty, ptr<int64_t>(out),
                             max_out, ptr<int64_t>(out_len), ptr<void>(workspace), workspace_bytes, ptr<void>(stream)));
        },
        py::arg("lp"), py::arg("dtype"), py::arg("sB"), py::arg("sT"), py::arg("sV"), py::arg("x_len"), py::arg("B"),
        py::arg("T"), py::arg("V"), py::arg("blank"), py::arg("beam_width"), py::arg("space_id"), py::arg("lm"),
        py::arg("lmwt"), py::arg("wip"), py::arg("oov_penalty"), py::arg("out"), py::arg("max_out"),
        py::arg("out_len"), py::arg("workspace"), py::arg("workspace_bytes"), py::arg("stream"));

  m.def("ctc_align_workspace_bytes",
        [](int B, int T, int V, int Smax, bool is_ctc) { return e2e_ctc_align_workspace_bytes(B, T, V, Smax, is_ctc ? 1 : 0); });

  m.def("ctc_align",
        [](uintptr_t lp, int dtype, int64_t sB, int64_t sT, int64_t sV, uintptr_t targets, int64_t tgt_stride,
           uintptr_t x_len, uintptr_t t_len, int B, int T, int V, int Smax, int blank, bool is_ctc, uintptr_t out,
           int64_t pad_value, uintptr_t workspace, size_t workspace_bytes, uintptr_t stream) {
          check(e2e_ctc_align(ptr<const void>(lp), dtype, sB, sT, sV, ptr<const int64_t>(targets), tgt_stride,
                              ptr<const int64_t>(x_len), ptr<const int64_t>(t_len), B, T, V, Smax, blank,
                              is_ctc ? 1 : 0, ptr<int64_t>(out), pad_value, ptr<void>(workspace), workspace_bytes,
                              ptr<void>(stream)));
        },
        py::arg("lp"), py::arg("dtype"), py::arg("sB"), py::arg("sT"), py::arg("sV"), py::arg("targets"),
        py::arg("tgt_stride"), py::arg("x_len"), py::arg("t_len"), py::arg("B"), py::arg("T"), py::arg("V"),
        py::arg("Smax"), py::arg("blank"), py::arg("is_ctc"), py::arg("out"), py::arg("pad_value"),
        py::arg("workspace"), py::arg("workspace_bytes"), py::arg("stream"));

  py::class_<LanguageModel>(m, "LanguageModel")
      .def(py::init<const std::string&, const std::vector<std::string>&, bool>(), py::arg("path"), py::arg("labels"),
           py::arg("case_sensitive"))
      .def_property_readonly("handle", &LanguageModel::handle)
      .def("order", &LanguageModel::order)
      .def("device", &LanguageModel::device)
      .def("word_index", &LanguageModel::word_index, py::arg("word"))
      .def("score", &LanguageModel::score, py::arg("ctx"), py::arg("word"));
}
