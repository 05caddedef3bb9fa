// vmm_ops.cpp — the Python module `kvcached_amd.vmm_ops` (aliased as `kvcached.vmm_ops`).
//
// Re-creates, name for name and default for default, the surface the reference defines at
// csrc/torch_bindings.cpp:182-258 — six free functions plus the PageAllocator and InternalPage
// classes — as a thin pybind11 layer over the C ABI in include/kvcached_amd.h. Nothing in this
// file touches HIP; torch is used for exactly one thing, wrapping the reserved VA ranges as
// non-owning tensors (at::from_blob, like csrc/ftensor.cpp:71-75).
//
// Threading differs from the reference on purpose: every call that can block or reach the
// driver releases the GIL (the reference holds it through alloc_page(), which can deadlock
// against its own prealloc thread — SURVEY §5), and Python callbacks re-acquire it in the
// trampolines below.
#include <pybind11/functional.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <ATen/ATen.h>
#include <torch/csrc/utils/pybind.h>

#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "kvcached_amd.h"

namespace py = pybind11;

namespace {

[[noreturn]] void raise_last(int code) {
  std::string msg = kvc_last_error();
  if (msg.empty()) msg = "kvcached_amd error " + std::to_string(code);
  throw std::runtime_error(msg);
}
inline void check(int rc) {
  if (rc < 0) raise_last(rc);
}

// dtype is chosen by element size only, as the reference does (csrc/inc/impl/torch_utils.ipp:32-46)
c10::ScalarType dtype_from_size(size_t n) {
  switch (n) {
  case 1: return c10::ScalarType::Char;
  case 2: return c10::ScalarType::Short;
  case 4: return c10::ScalarType::Int;
  case 8: return c10::ScalarType::Long;
  default: throw std::runtime_error("Unsupported dtype size: " + std::to_string(n));
  }
}

void init_kvcached(const std::string &dev_str, size_t page_size, bool contiguous_layout) {
  py::gil_scoped_release nogil;
  check(kvc_init(dev_str.c_str(), page_size, contiguous_layout ? 1 : 0));
}
void shutdown_kvcached() {
  py::gil_scoped_release nogil;
  check(kvc_shutdown());
}

std::vector<at::Tensor> create_kv_tensors(size_t size, size_t dtype_size, const std::string &dev_str,
                                          int64_t num_layers, int64_t num_kv_buffers, int64_t group_id,
                                          bool unified_pool) {
  py::gil_scoped_release nogil;
  const auto dtype = dtype_from_size(dtype_size);
  std::vector<void *> ptrs(num_layers > 0 ? (size_t)num_layers : 1);
  std::vector<size_t> nbytes(ptrs.size());
  int64_t count = (int64_t)ptrs.size();
  check(kvc_create_kv_tensors(size, dtype_size, dev_str.c_str(), num_layers, num_kv_buffers, group_id,
                              unified_pool ? 1 : 0, ptrs.data(), nbytes.data(), &count));
  int is_gpu = 0, index = 0;
  check(kvc_get_device(&is_gpu, &index));
  const c10::Device dev = is_gpu ? c10::Device(c10::kCUDA, (c10::DeviceIndex)index) : c10::Device(c10::kCPU);
  auto opts = at::TensorOptions().dtype(dtype).device(dev).requires_grad(false);
  std::vector<at::Tensor> out;
  out.reserve((size_t)count);
  for (int64_t i = 0; i < count; ++i) {
    // target_device is given explicitly: unbacked VA must not be probed for its owner
    out.push_back(at::for_blob(ptrs[i], {(int64_t)(nbytes[i] / dtype_size)}).options(opts).target_device(dev).make_tensor());
  }
  return out;
}

bool kv_tensors_created(int64_t group_id) {
  py::gil_scoped_release nogil;
  int rc = kvc_kv_tensors_created(group_id);
  check(rc);
  return rc == 1;
}
bool map_to_kv_tensors(const std::vector<int64_t> &offsets, int64_t group_id) {
  py::gil_scoped_release nogil;
  int rc = kvc_map_to_kv_tensors(offsets.data(), offsets.size(), group_id);
  if (rc == KVC_E_NOT_CREATED) return false; // the reference logs and returns False
  check(rc);
  return true;
}
bool unmap_from_kv_tensors(const std::vector<int64_t> &offsets, int64_t group_id) {
  py::gil_scoped_release nogil;
  int rc = kvc_unmap_from_kv_tensors(offsets.data(), offsets.size(), group_id);
  if (rc == KVC_E_NOT_CREATED) return false;
  check(rc);
  return true;
}

// ---------------------------------------------------------------- block ids <-> token indices
// Additions to the reference surface (the fused replacement of the torch/Triton glue in
// kvcached/integration/sglang/patches.py:186-288). Device arrays are raw addresses (tensor.data_ptr()), the
// stream is torch's current stream handle; the id lists are read with the raw CPython API because these calls
// sit on the scheduler's critical path (a ctypes array of 1024 ids costs 30 us to build, this 3 us).
namespace {
std::vector<int64_t> ids_from_sequence(py::handle seq) {
  PyObject *fast = PySequence_Fast(seq.ptr(), "block ids must be a list or tuple of int");
  if (!fast) throw py::error_already_set();
  const Py_ssize_t n = PySequence_Fast_GET_SIZE(fast);
  std::vector<int64_t> out((size_t)n);
  PyObject **items = PySequence_Fast_ITEMS(fast);
  for (Py_ssize_t i = 0; i < n; ++i) {
    const long long v = PyLong_AsLongLong(items[i]);
    if (v == -1 && PyErr_Occurred()) {
      Py_DECREF(fast);
      throw py::error_already_set();
    }
    out[(size_t)i] = v;
  }
  Py_DECREF(fast);
  return out;
}
template <class T> T *as_ptr(uintptr_t a) { return reinterpret_cast<T *>(a); }
} // namespace

void expand_block_ids(py::handle block_ids, int64_t tokens_per_block, uintptr_t out_ptr, uintptr_t stream) {
  auto ids = ids_from_sequence(block_ids);
  check(kvc_expand_block_ids(ids.data(), ids.size(), tokens_per_block, as_ptr<int64_t>(out_ptr), as_ptr<void>(stream)));
}
void alloc_extend_indices(uintptr_t prefix_lens, uintptr_t seq_lens, uintptr_t last_loc, size_t bs, py::handle new_block_ids,
                          int64_t tokens_per_block, uintptr_t out_ptr, size_t extend_num_tokens, uintptr_t stream) {
  auto ids = ids_from_sequence(new_block_ids);
  check(kvc_alloc_extend_indices(as_ptr<const int64_t>(prefix_lens), as_ptr<const int64_t>(seq_lens),
                                 as_ptr<const int64_t>(last_loc), bs, ids.data(), ids.size(), tokens_per_block,
                                 as_ptr<int64_t>(out_ptr), extend_num_tokens, as_ptr<void>(stream)));
}
void alloc_decode_indices(uintptr_t seq_lens, uintptr_t last_loc, size_t bs, py::handle new_block_ids,
                          int64_t tokens_per_block, uintptr_t out_ptr, uintptr_t stream) {
  auto ids = ids_from_sequence(new_block_ids);
  check(kvc_alloc_decode_indices(as_ptr<const int64_t>(seq_lens), as_ptr<const int64_t>(last_loc), bs, ids.data(), ids.size(),
                                 tokens_per_block, as_ptr<int64_t>(out_ptr), as_ptr<void>(stream)));
}
py::list unique_block_ids(uintptr_t token_indices, size_t n, int64_t tokens_per_block, int64_t num_blocks, uintptr_t stream) {
  std::vector<int64_t> out(std::max<size_t>(1, std::min<size_t>(n, (size_t)std::max<int64_t>(0, num_blocks))));
  int64_t cnt;
  {
    py::gil_scoped_release nogil; // blocks until the sweep kernel's result is on the host
    cnt = kvc_unique_block_ids(as_ptr<const int64_t>(token_indices), n, tokens_per_block, num_blocks, out.data(), out.size(),
                               as_ptr<void>(stream));
  }
  if (cnt < 0) raise_last((int)cnt);
  PyObject *l = PyList_New((Py_ssize_t)cnt);
  if (!l) throw py::error_already_set();
  for (int64_t i = 0; i < cnt; ++i) PyList_SET_ITEM(l, (Py_ssize_t)i, PyLong_FromLongLong(out[(size_t)i]));
  return py::reinterpret_steal<py::list>(l);
}

// ---------------------------------------------------------------- InternalPage
class PyInternalPage {
public:
  PyInternalPage(int64_t page_id, int64_t page_size) : p_(kvc_page_new(page_id, page_size)) {}
  ~PyInternalPage() { kvc_page_delete(p_); }
  PyInternalPage(const PyInternalPage &) = delete;
  int64_t page_id() const { return kvc_page_id(p_); }
  int64_t page_size() const { return kvc_page_size(p_); }
  void init(int64_t block_mem_size) { kvc_page_init(p_, block_mem_size); }
  std::vector<int64_t> alloc(int64_t num_blocks) {
    std::vector<int64_t> out((size_t)std::max<int64_t>(num_blocks, 0));
    int64_t n = kvc_page_alloc(p_, num_blocks, out.data());
    if (n < 0) raise_last((int)n);
    return out;
  }
  void free(int64_t block_id) { kvc_page_free(p_, block_id); }
  void free_batch(const std::vector<int64_t> &ids) { kvc_page_free_batch(p_, ids.data(), ids.size()); }
  bool empty() const { return kvc_page_empty(p_) != 0; }
  bool full() const { return kvc_page_full(p_) != 0; }
  int64_t num_free_blocks() const { return kvc_page_num_free_blocks(p_); }
  std::vector<int64_t> get_free_blocks() const {
    std::vector<int64_t> out((size_t)kvc_page_num_free_blocks(p_));
    kvc_page_get_free_blocks(p_, out.data(), (int64_t)out.size());
    return out;
  }
  static std::pair<int64_t, int64_t> get_block_range(int64_t pid, int64_t P, int64_t B) {
    int64_t s = 0, e = 0;
    kvc_page_get_block_range(pid, P, B, &s, &e);
    return {s, e};
  }
  static int64_t get_num_blocks(int64_t P, int64_t B) { return kvc_page_get_num_blocks(P, B); }

private:
  kvc_page_t *p_;
};

// ---------------------------------------------------------------- PageAllocator
class PyPageAllocator {
public:
  PyPageAllocator(int64_t num_layers, int64_t mem_size_per_layer, int64_t page_size, int64_t world_size,
                  int64_t pp_rank, bool async_sched, bool contiguous_layout, bool enable_page_prealloc,
                  int64_t num_kv_buffers, int64_t group_id, const std::string &ipc_name)
      : page_size_(page_size) {
    pa_ = kvc_pa_new(num_layers, mem_size_per_layer, page_size, world_size, pp_rank, async_sched, contiguous_layout,
                     enable_page_prealloc, num_kv_buffers, group_id, ipc_name.c_str());
    if (!pa_) raise_last(KVC_E_RUNTIME);
  }
  ~PyPageAllocator() {
    {
      py::gil_scoped_release nogil; // joins the background threads, which may be inside a callback
      kvc_pa_delete(pa_);
    }
    // the py::function members are destroyed with the GIL held (we are in a Python-called dtor)
  }
  PyPageAllocator(const PyPageAllocator &) = delete;

  void start_prealloc_thread() {
    py::gil_scoped_release nogil;
    check(kvc_pa_start_prealloc_thread(pa_));
  }
  void stop_prealloc_thread() {
    py::gil_scoped_release nogil;
    check(kvc_pa_stop_prealloc_thread(pa_));
  }
  std::shared_ptr<PyInternalPage> alloc_page() {
    int64_t pid;
    {
      py::gil_scoped_release nogil;
      pid = kvc_pa_alloc_page(pa_);
    }
    if (pid < 0) raise_last((int)pid);
    return std::make_shared<PyInternalPage>(pid, page_size_);
  }
  // addition: n pages, one map call for those that need backing (see kvc_pa_alloc_pages)
  std::vector<std::shared_ptr<PyInternalPage>> alloc_pages(int64_t n) {
    std::vector<int64_t> ids((size_t)std::max<int64_t>(n, 1));
    int64_t k;
    {
      py::gil_scoped_release nogil;
      k = kvc_pa_alloc_pages(pa_, n, ids.data());
    }
    if (k < 0) raise_last((int)k);
    std::vector<std::shared_ptr<PyInternalPage>> out;
    for (int64_t i = 0; i < k; ++i) out.push_back(std::make_shared<PyInternalPage>(ids[(size_t)i], page_size_));
    return out;
  }
  void free_page(int64_t page_id) {
    py::gil_scoped_release nogil;
    check(kvc_pa_free_page(pa_, page_id));
  }
  void free_pages(const std::vector<int64_t> &ids) {
    py::gil_scoped_release nogil;
    check(kvc_pa_free_pages(pa_, ids.data(), ids.size()));
  }
  bool resize(int64_t new_mem_size) {
    py::gil_scoped_release nogil;
    int rc = kvc_pa_resize(pa_, new_mem_size);
    check(rc);
    return rc == 1;
  }
  void trim() {
    py::gil_scoped_release nogil;
    check(kvc_pa_trim(pa_));
  }
  void reset_free_page_order() { check(kvc_pa_reset_free_page_order(pa_)); }
  int64_t get_num_free_pages() const { return kvc_pa_get_num_free_pages(pa_); }
  int64_t get_num_inuse_pages() const { return kvc_pa_get_num_inuse_pages(pa_); }
  int64_t get_num_total_pages() const { return kvc_pa_get_num_total_pages(pa_); }
  int64_t get_num_reserved_pages() const { return kvc_pa_get_num_reserved_pages(pa_); }
  int64_t get_avail_physical_pages() const {
    int64_t n = kvc_pa_get_avail_physical_pages(pa_);
    if (n < 0) raise_last((int)n);
    return n;
  }
  int64_t check_and_get_resize_target(int64_t cur) const { return kvc_pa_check_and_get_resize_target(pa_, cur); }
  int64_t get_resize_target() const { return kvc_pa_get_resize_target(pa_); }
  int64_t get_page_id(int64_t block_id, int64_t block_mem_size) const {
    return kvc_pa_get_page_id(pa_, block_id, block_mem_size);
  }
  // Takes the Python list as it is (no std::vector round trip) and builds the result with the raw
  // CPython API: this call sits on every KVCacheManager.free().
  py::dict group_indices_by_page(py::list indices, int64_t block_mem_size) const {
    const size_t n = indices.size();
    std::vector<int64_t> in(n ? n : 1), keys(n ? n : 1), counts(n ? n : 1), values(n ? n : 1);
    for (size_t i = 0; i < n; ++i) {
      const long long v = PyLong_AsLongLong(PyList_GET_ITEM(indices.ptr(), (Py_ssize_t)i));
      if (v == -1 && PyErr_Occurred()) throw py::error_already_set();
      in[i] = v;
    }
    int64_t k = kvc_pa_group_indices_by_page(pa_, in.data(), n, block_mem_size, keys.data(), counts.data(), values.data());
    if (k < 0) raise_last((int)k);
    py::dict d; // built in the C++ map's iteration order, like pybind11's stl caster does for the reference
    size_t w = 0;
    for (int64_t i = 0; i < k; ++i) {
      PyObject *l = PyList_New((Py_ssize_t)counts[i]);
      if (!l) throw py::error_already_set();
      for (int64_t j = 0; j < counts[i]; ++j) PyList_SET_ITEM(l, (Py_ssize_t)j, PyLong_FromLongLong(values[w++]));
      PyObject *key = PyLong_FromLongLong(keys[i]);
      const int rc = PyDict_SetItem(d.ptr(), key, l);
      Py_DECREF(key);
      Py_DECREF(l);
      if (rc != 0) throw py::error_already_set();
    }
    return d;
  }

  void set_broadcast_map_callback(py::object fn) {
    map_cb_ = std::move(fn);
    check(kvc_pa_set_broadcast_map_callback(pa_, map_cb_.is_none() ? nullptr : &PyPageAllocator::tramp_map, this));
  }
  void set_broadcast_unmap_callback(py::object fn) {
    unmap_cb_ = std::move(fn);
    check(kvc_pa_set_broadcast_unmap_callback(pa_, unmap_cb_.is_none() ? nullptr : &PyPageAllocator::tramp_unmap, this));
  }
  void set_should_use_worker_ipc_callback(py::object fn) {
    ipc_cb_ = std::move(fn);
    check(kvc_pa_set_should_use_worker_ipc_callback(pa_, ipc_cb_.is_none() ? nullptr : &PyPageAllocator::tramp_ipc, this));
  }
  std::vector<int64_t> _page_list(int which) const {
    std::vector<int64_t> out((size_t)kvc_pa_get_page_list(pa_, which, nullptr, 0));
    kvc_pa_get_page_list(pa_, which, out.data(), (int64_t)out.size());
    return out;
  }
  std::string _ipc_name() const { return kvc_pa_ipc_name(pa_); }

private:
  static int call_broadcast(py::object &fn, int64_t ws, const int64_t *off, size_t n) {
    py::gil_scoped_acquire gil;
    try {
      py::list l(n);
      for (size_t i = 0; i < n; ++i) l[i] = py::int_(off[i]);
      fn(ws, l);
      return 0;
    } catch (py::error_already_set &e) {
      fprintf(stderr, "[kvcached_amd] broadcast callback raised: %s\n", e.what());
      e.discard_as_unraisable("kvcached_amd broadcast callback");
      return -1;
    }
  }
  static int tramp_map(void *self, int64_t ws, const int64_t *off, size_t n) {
    return call_broadcast(static_cast<PyPageAllocator *>(self)->map_cb_, ws, off, n);
  }
  static int tramp_unmap(void *self, int64_t ws, const int64_t *off, size_t n) {
    return call_broadcast(static_cast<PyPageAllocator *>(self)->unmap_cb_, ws, off, n);
  }
  static int tramp_ipc(void *self) {
    py::gil_scoped_acquire gil;
    try {
      return static_cast<PyPageAllocator *>(self)->ipc_cb_().cast<bool>() ? 1 : 0;
    } catch (py::error_already_set &e) {
      e.discard_as_unraisable("kvcached_amd should_use_worker_ipc callback");
      return 0;
    }
  }

  kvc_page_allocator_t *pa_ = nullptr;
  int64_t page_size_;
  py::object map_cb_ = py::none(), unmap_cb_ = py::none(), ipc_cb_ = py::none();
};

} // namespace

PYBIND11_MODULE(TORCH_EXTENSION_NAME, m) {
  m.doc() = "kvcached VMM plugin (MI355X-native build)";

  m.def("init_kvcached", &init_kvcached, "Initialize kvcached", py::arg("dev_str"), py::arg("page_size") = 0,
        py::arg("contiguous_layout") = true);
  m.def("shutdown_kvcached", &shutdown_kvcached, "Shutdown kvcached");
  m.def("create_kv_tensors", &create_kv_tensors, "create_kv_tensors", py::arg("size"), py::arg("dtype_size"),
        py::arg("dev_str"), py::arg("num_layers"), py::arg("num_kv_buffers") = 2, py::arg("group_id") = 0,
        py::arg("unified_pool") = false);
  m.def("kv_tensors_created", &kv_tensors_created, "kv_tensors_created", py::arg("group_id") = 0);
  m.def("map_to_kv_tensors", &map_to_kv_tensors, "map_to_kv_tensors", py::arg("offsets"), py::arg("group_id") = 0);
  m.def("unmap_from_kv_tensors", &unmap_from_kv_tensors, "unmap_from_kv_tensors", py::arg("offsets"),
        py::arg("group_id") = 0);

  // additions (no reference counterpart): fused block id <-> token index glue for the SGLang allocators
  m.def("expand_block_ids", &expand_block_ids, py::arg("block_ids"), py::arg("tokens_per_block"), py::arg("out_ptr"),
        py::arg("stream") = 0);
  m.def("alloc_extend_indices", &alloc_extend_indices, py::arg("prefix_lens_ptr"), py::arg("seq_lens_ptr"),
        py::arg("last_loc_ptr"), py::arg("bs"), py::arg("new_block_ids"), py::arg("tokens_per_block"), py::arg("out_ptr"),
        py::arg("extend_num_tokens"), py::arg("stream") = 0);
  m.def("alloc_decode_indices", &alloc_decode_indices, py::arg("seq_lens_ptr"), py::arg("last_loc_ptr"), py::arg("bs"),
        py::arg("new_block_ids"), py::arg("tokens_per_block"), py::arg("out_ptr"), py::arg("stream") = 0);
  m.def("unique_block_ids", &unique_block_ids, py::arg("token_indices_ptr"), py::arg("n"), py::arg("tokens_per_block"),
        py::arg("num_blocks"), py::arg("stream") = 0);

  py::class_<PyPageAllocator, std::shared_ptr<PyPageAllocator>>(m, "PageAllocator")
      .def(py::init<int64_t, int64_t, int64_t, int64_t, int64_t, bool, bool, bool, int64_t, int64_t, const std::string &>(),
           py::arg("num_layers"), py::arg("mem_size_per_layer"), py::arg("page_size"), py::arg("world_size") = 1,
           py::arg("pp_rank") = 0, py::arg("async_sched") = false, py::arg("contiguous_layout") = true,
           py::arg("enable_page_prealloc") = true, py::arg("num_kv_buffers") = 2, py::arg("group_id") = 0,
           py::arg("ipc_name") = "")
      .def("start_prealloc_thread", &PyPageAllocator::start_prealloc_thread)
      .def("stop_prealloc_thread", &PyPageAllocator::stop_prealloc_thread)
      .def("alloc_page", &PyPageAllocator::alloc_page)
      .def("alloc_pages", &PyPageAllocator::alloc_pages)
      .def("free_page", &PyPageAllocator::free_page)
      .def("free_pages", &PyPageAllocator::free_pages)
      .def("resize", &PyPageAllocator::resize)
      .def("trim", &PyPageAllocator::trim)
      .def("reset_free_page_order", &PyPageAllocator::reset_free_page_order)
      .def("get_num_free_pages", &PyPageAllocator::get_num_free_pages)
      .def("get_num_inuse_pages", &PyPageAllocator::get_num_inuse_pages)
      .def("get_num_total_pages", &PyPageAllocator::get_num_total_pages)
      .def("get_num_reserved_pages", &PyPageAllocator::get_num_reserved_pages)
      .def("get_avail_physical_pages", &PyPageAllocator::get_avail_physical_pages)
      .def("check_and_get_resize_target", &PyPageAllocator::check_and_get_resize_target)
      .def("get_resize_target", &PyPageAllocator::get_resize_target)
      .def("get_page_id", &PyPageAllocator::get_page_id)
      .def("group_indices_by_page", &PyPageAllocator::group_indices_by_page)
      .def("set_broadcast_map_callback", &PyPageAllocator::set_broadcast_map_callback)
      .def("set_broadcast_unmap_callback", &PyPageAllocator::set_broadcast_unmap_callback)
      .def("set_should_use_worker_ipc_callback", &PyPageAllocator::set_should_use_worker_ipc_callback)
      // additions (underscore-prefixed: not part of the reference surface)
      .def("_page_list", &PyPageAllocator::_page_list)
      .def("_ipc_name", &PyPageAllocator::_ipc_name);

  py::class_<PyInternalPage, std::shared_ptr<PyInternalPage>>(m, "InternalPage")
      .def(py::init<int64_t, int64_t>(), py::arg("page_id"), py::arg("page_size"))
      .def_property_readonly("page_id", &PyInternalPage::page_id)
      .def_property_readonly("page_size", &PyInternalPage::page_size)
      .def("init", &PyInternalPage::init)
      .def("alloc", &PyInternalPage::alloc)
      .def("free", &PyInternalPage::free)
      .def("free_batch", &PyInternalPage::free_batch)
      .def("empty", &PyInternalPage::empty)
      .def("full", &PyInternalPage::full)
      .def("num_free_blocks", &PyInternalPage::num_free_blocks)
      .def("get_free_blocks", &PyInternalPage::get_free_blocks)
      .def_static("get_block_range", &PyInternalPage::get_block_range)
      .def_static("get_num_blocks", &PyInternalPage::get_num_blocks);
}
