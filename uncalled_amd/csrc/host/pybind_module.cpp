// pybind11 module `_uncalled_amd`: the subset of the reference's `_uncalled` (src/pybinder.cpp:14-91) that
// `uncalled map` touches -- Conf, MapPool, Paf -- with the same names, properties and methods.
#include <pybind11/numpy.h>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include "unc_realtime.hpp"

namespace py = pybind11;
using namespace unc_host;

// tests link these sources a second time against the lanesim build of the C ABI and load both modules into one process:
// that copy registers its types module-locally
#ifdef UNC_PYBIND_LOCAL
#define UNC_ML , py::module_local()
#else
#define UNC_ML
#endif

PYBIND11_MODULE(_uncalled_amd, m) {
    m.doc() = "MI355X-native UNCALLED map path: Conf / MapPool / Paf over libuncalled_hip.so";

    py::class_<Conf>(m, "Conf" UNC_ML)
        .def(py::init<>())
#define PRP(P) .def_readwrite(#P, &Conf::P)
        PRP(threads) PRP(bwa_prefix) PRP(idx_preset) PRP(model_path) PRP(max_events) PRP(seed_len) PRP(chunk_time) PRP(fast5_list)
        PRP(read_list) PRP(max_reads) PRP(max_buffer) PRP(num_channels) PRP(max_chunks) PRP(sample_rate) PRP(device) PRP(batch_reads)
        PRP(realtime_mode) PRP(active_chs) PRP(duration) PRP(max_active_reads);
#undef PRP

    py::class_<Paf> paf(m, "Paf" UNC_ML);
    paf.def(py::init<>())
        .def("print_paf", &Paf::print_paf)
        .def("__str__", &Paf::str)
        .def("is_mapped", &Paf::is_mapped)
        .def("is_ended", &Paf::is_ended)
        .def("set_int", &Paf::set_int)
        .def("set_float", &Paf::set_float)
        .def("set_str", &Paf::set_str);
    py::enum_<Paf::Tag>(paf, "Tag" UNC_ML)
        .value("MAP_TIME", Paf::MAP_TIME).value("EJECT", Paf::EJECT).value("IN_SCAN", Paf::IN_SCAN).value("ENDED", Paf::ENDED)
        .value("KEEP", Paf::KEEP).value("DELAY", Paf::DELAY).value("WAIT_TIME", Paf::WAIT_TIME).value("CHANNEL", Paf::CHANNEL)
        .value("READ_START", Paf::READ_START)
        .export_values();

    // ReadBuffer as Fast5Reader::pop_read returns it (read_buffer.hpp:182-198): id / start / channel / raw, with the
    // int16 samples and the calibration triple beside the calibrated floats
    py::class_<RawRead>(m, "ReadBuffer" UNC_ML)
        .def("empty", [](const RawRead &r) { return r.signal.empty(); })
        .def("size", [](const RawRead &r) { return r.signal.size(); })
        .def_property_readonly("id", [](const RawRead &r) { return r.id; })
        .def_property_readonly("start", [](const RawRead &r) { return r.start_sample; })
        .def_property_readonly("number", [](const RawRead &r) { return r.number; })
        .def_property_readonly("channel", [](const RawRead &r) { return (uint16_t)(r.channel_idx + 1); })
        .def_property_readonly("calibration", [](const RawRead &r) { return py::make_tuple(r.calib.range, r.calib.offset, r.calib.digitisation); })
        .def_property_readonly("raw_i16", [](const RawRead &r) { return r.signal; })
        .def_property_readonly("raw", [](const RawRead &r) {   // fast5.hpp raw_samples_to_float arithmetic
            std::vector<float> out(r.signal.size());
            for (size_t i = 0; i < out.size(); ++i) {
                const float t1 = (float)(int)(uint16_t)r.signal[i] + r.calib.offset;
                const float t2 = r.calib.range * t1;
                out[i] = t2 / r.calib.digitisation;
            }
            return out;
        });

    m.def("write_fast5",
          [](const std::string &path, const std::vector<py::dict> &reads, bool multi, float sample_rate) {
              std::vector<RawRead> rs;
              for (const py::dict &d : reads) {
                  RawRead r;
                  r.id = d["id"].cast<std::string>();
                  r.channel_idx = (uint16_t)(d["channel"].cast<int>() - 1);
                  r.number = d.contains("number") ? d["number"].cast<uint32_t>() : 0;
                  r.start_sample = d.contains("start") ? d["start"].cast<uint64_t>() : 0;
                  r.calib.range = d["range"].cast<float>();
                  r.calib.offset = d["offset"].cast<float>();
                  r.calib.digitisation = d["digitisation"].cast<float>();
                  if (py::isinstance<py::array>(d["signal"])) {          // numpy int16: one memcpy instead of a cast per sample
                      auto a = py::array_t<int16_t, py::array::c_style | py::array::forcecast>::ensure(d["signal"]);
                      r.signal.assign(a.data(), a.data() + a.size());
                  } else {
                      r.signal = d["signal"].cast<std::vector<int16_t>>();
                  }
                  rs.push_back(std::move(r));
              }
              return write_fast5(path, rs, multi, sample_rate);
          },
          py::arg("path"), py::arg("reads"), py::arg("multi") = true, py::arg("sample_rate") = 4000.0f);

    py::class_<Fast5Reader>(m, "Fast5Reader" UNC_ML)
        .def(py::init<const Conf &>())
        .def(py::init([](const std::string &fast5_list, const std::string &read_list, uint32_t max_reads, uint32_t max_buffer) {
            Conf c; c.fast5_list = fast5_list; c.read_list = read_list; c.max_reads = max_reads; c.max_buffer = max_buffer;
            return new Fast5Reader(c);
        }))
        .def("add_fast5", &Fast5Reader::add_fast5)
        .def("load_fast5_list", &Fast5Reader::load_fast5_list)
        .def("add_read", &Fast5Reader::add_read)
        .def("load_read_list", &Fast5Reader::load_read_list)
        .def("pop_read", &Fast5Reader::pop_read)
        .def("buffer_size", &Fast5Reader::buffered)
        .def("fill_buffer", &Fast5Reader::fill_buffer)
        .def("all_buffered", &Fast5Reader::all_buffered)
        .def("empty", &Fast5Reader::empty);

    // ---- realtime path: Chunk / RealtimePool / ClientSim with the reference's names (pybinder.cpp:33-47,
    //      chunk.hpp:47-59, realtime_pool.hpp:63-70, client_sim.hpp:57-75)
    py::class_<Chunk>(m, "Chunk" UNC_ML)
        .def(py::init<>())
        .def(py::init([](const std::string &id, uint16_t channel, uint32_t number, uint64_t start, const std::string &dtype, const py::bytes &raw) {
            return new Chunk(id, channel, number, start, dtype, std::string(raw));
        }))
        .def(py::init<const std::string &, uint16_t, uint32_t, uint64_t, const std::vector<float> &, uint32_t, uint32_t>())
        .def("pop", [](Chunk &c) { std::vector<float> v; c.pop(v); return v; })
        .def("swap", &Chunk::swap)
        .def("empty", &Chunk::empty)
        .def("print", &Chunk::print)
        .def("size", &Chunk::size)
        .def_property_readonly("channel", &Chunk::get_channel)
        .def_property_readonly("number", &Chunk::get_number)
        .def_property_readonly("id", &Chunk::get_id)
        .def_property_readonly("start", &Chunk::get_start);

    py::class_<RealtimePool> rp(m, "RealtimePool" UNC_ML);
    rp.def(py::init<const Conf &>())
        .def("add_chunk", &RealtimePool::add_chunk)
        .def("try_add_chunk", &RealtimePool::try_add_chunk)
        .def("end_read", &RealtimePool::end_read)
        .def("update", &RealtimePool::update)
        .def("all_finished", &RealtimePool::all_finished)
        .def("stop_all", &RealtimePool::stop_all)
        .def("active_count", &RealtimePool::active_count)
        .def("last_round_ms", &RealtimePool::last_round_ms)
        .def("refused_chunks", &RealtimePool::refused_chunks);
    py::enum_<RealtimePool::Mode>(rp, "RealtimeMode" UNC_ML).value("DEPLETE", RealtimePool::DEPLETE).value("ENRICH", RealtimePool::ENRICH).export_values();
    py::enum_<RealtimePool::ActiveChs>(rp, "ActiveChs" UNC_ML)
        .value("FULL", RealtimePool::FULL).value("EVEN", RealtimePool::EVEN).value("ODD", RealtimePool::ODD).export_values();

    py::class_<ClientSim>(m, "ClientSim" UNC_ML)
        .def(py::init<const Conf &>())
        .def("add_fast5", &ClientSim::add_fast5)
        .def("load_fast5s", &ClientSim::load_fast5s)
        .def("run", &ClientSim::run)
        .def("get_read_chunks", &ClientSim::get_read_chunks)
        .def("stop_receiving_read", &ClientSim::stop_receiving_read)
        .def("unblock_read", &ClientSim::unblock_read)
        .def("get_runtime", &ClientSim::get_runtime)
        .def_property_readonly("is_running", &ClientSim::is_running);

    py::class_<MapPool>(m, "MapPool" UNC_ML)
        .def(py::init<const Conf &>())
        .def("update", &MapPool::update)
        .def("running", &MapPool::running)
        .def("add_fast5", &MapPool::add_fast5)
        .def("batch_reads", &MapPool::batch_reads)
        .def("stop", &MapPool::stop);
}
