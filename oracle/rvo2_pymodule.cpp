// ORACLE — TEST INFRASTRUCTURE ONLY (see rvo2_oracle.hpp header).
// Python module `rvo2` exposing the subset of the Python-RVO2 API that the reference uses
// (crowd_sim/envs/policy/orca.py:95-129, crowd_sim/envs/crowd_sim.py:221-245), backed by the float32
// restatement in rvo2_oracle.cpp.  With PYTHONPATH=oracle/_build:oracle/shims:/root/reference the
// UNMODIFIED reference Python runs on top of it; that combination generates tests/golden/.
// Every double coming from Python is narrowed to float here, exactly as the Cython wrapper does.
#include <pybind11/pybind11.h>

#include <memory>
#include <stdexcept>

#include "rvo2_oracle.hpp"

namespace py = pybind11;
using rvo2_oracle::Simulator;

namespace {

std::pair<float, float> xy(const py::handle& obj) {
    py::sequence s = py::reinterpret_borrow<py::sequence>(obj);
    if (py::len(s) != 2) throw std::invalid_argument("expected a 2-sequence");
    return {static_cast<float>(s[0].cast<double>()), static_cast<float>(s[1].cast<double>())};
}

class PySim {
public:
    PySim(double timeStep, double neighborDist, std::size_t maxNeighbors, double timeHorizon,
          double timeHorizonObst, double radius, double maxSpeed, py::object velocity)
        : nd_(static_cast<float>(neighborDist)),
          mn_(maxNeighbors),
          th_(static_cast<float>(timeHorizon)),
          tho_(static_cast<float>(timeHorizonObst)),
          r_(static_cast<float>(radius)),
          ms_(static_cast<float>(maxSpeed)) {
        (void)velocity;
        sim_ = std::make_unique<Simulator>(static_cast<float>(timeStep), nd_, mn_, th_, tho_, r_, ms_);
    }

    std::size_t addAgent(py::object pos, py::object neighborDist, py::object maxNeighbors,
                         py::object timeHorizon, py::object timeHorizonObst, py::object radius,
                         py::object maxSpeed, py::object velocity) {
        const auto p = xy(pos);
        const float nd = neighborDist.is_none() ? nd_ : static_cast<float>(neighborDist.cast<double>());
        const std::size_t mn = maxNeighbors.is_none() ? mn_ : maxNeighbors.cast<std::size_t>();
        const float th = timeHorizon.is_none() ? th_ : static_cast<float>(timeHorizon.cast<double>());
        const float tho =
            timeHorizonObst.is_none() ? tho_ : static_cast<float>(timeHorizonObst.cast<double>());
        const float r = radius.is_none() ? r_ : static_cast<float>(radius.cast<double>());
        const float ms = maxSpeed.is_none() ? ms_ : static_cast<float>(maxSpeed.cast<double>());
        std::pair<float, float> v{0.0f, 0.0f};
        if (!velocity.is_none()) v = xy(velocity);
        return sim_->addAgent(p.first, p.second, nd, mn, th, tho, r, ms, v.first, v.second);
    }

    void check(std::size_t i) const {
        if (i >= sim_->numAgents()) throw std::out_of_range("agent index out of range");
    }
    void setAgentPosition(std::size_t i, py::object p) {
        check(i);
        const auto v = xy(p);
        sim_->setPosition(i, v.first, v.second);
    }
    void setAgentVelocity(std::size_t i, py::object p) {
        check(i);
        const auto v = xy(p);
        sim_->setVelocity(i, v.first, v.second);
    }
    void setAgentPrefVelocity(std::size_t i, py::object p) {
        check(i);
        const auto v = xy(p);
        sim_->setPrefVelocity(i, v.first, v.second);
    }
    py::tuple getAgentPosition(std::size_t i) const {
        check(i);
        const auto v = sim_->position(i);
        return py::make_tuple(static_cast<double>(v.x), static_cast<double>(v.y));
    }
    py::tuple getAgentVelocity(std::size_t i) const {
        check(i);
        const auto v = sim_->velocity(i);
        return py::make_tuple(static_cast<double>(v.x), static_cast<double>(v.y));
    }
    std::size_t getNumAgents() const { return sim_->numAgents(); }
    double getTimeStep() const { return sim_->timeStep(); }
    double getGlobalTime() const { return sim_->globalTime(); }
    void doStep() { sim_->doStep(); }

private:
    float nd_;
    std::size_t mn_;
    float th_, tho_, r_, ms_;
    std::unique_ptr<Simulator> sim_;
};

}  // namespace

PYBIND11_MODULE(rvo2, m) {
    m.doc() = "float32 restatement of RVO2 agent-agent ORCA (oracle; API subset of Python-RVO2)";
    py::class_<PySim>(m, "PyRVOSimulator")
        .def(py::init<double, double, std::size_t, double, double, double, double, py::object>(),
             py::arg("timeStep"), py::arg("neighborDist"), py::arg("maxNeighbors"),
             py::arg("timeHorizon"), py::arg("timeHorizonObst"), py::arg("radius"),
             py::arg("maxSpeed"), py::arg("velocity") = py::none())
        .def("addAgent", &PySim::addAgent, py::arg("pos"), py::arg("neighborDist") = py::none(),
             py::arg("maxNeighbors") = py::none(), py::arg("timeHorizon") = py::none(),
             py::arg("timeHorizonObst") = py::none(), py::arg("radius") = py::none(),
             py::arg("maxSpeed") = py::none(), py::arg("velocity") = py::none())
        .def("setAgentPosition", &PySim::setAgentPosition)
        .def("setAgentVelocity", &PySim::setAgentVelocity)
        .def("setAgentPrefVelocity", &PySim::setAgentPrefVelocity)
        .def("getAgentPosition", &PySim::getAgentPosition)
        .def("getAgentVelocity", &PySim::getAgentVelocity)
        .def("getNumAgents", &PySim::getNumAgents)
        .def("getTimeStep", &PySim::getTimeStep)
        .def("getGlobalTime", &PySim::getGlobalTime)
        .def("doStep", &PySim::doStep);
}
