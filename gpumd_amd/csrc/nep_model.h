// nep.txt -> host-side model description.
// Follows the file format consumed by the reference's NEP::NEP (src/force/nep.cu:100-377) and the
// nep3 conventions of its vendored NEP_CPU (tools/.../for_perioidc_table/nep.cpp:2568-2872).
#pragma once
#include <string>
#include <vector>

namespace nepmi {

constexpr int kMaxTypes = 94;

// one Tersoff-1989 parameter set + the derived constants of src/force/tersoff1989.cu:66-72
struct TersoffSet {
  double a = 0, b = 0, lambda = 0, mu = 0, beta = 0, n = 0, c = 0, d = 0, h = 0, r1 = 0, r2 = 0;
  double c2 = 0, d2 = 0, one_plus_c2overd2 = 0, pi_factor = 0, minus_half_over_n = 0;
};

struct NepModel {
  int kind = 0;    // 0: NEP, 1: Tersoff-1989 (BASELINE config 2)
  TersoffSet ters[3]; // type 0-0, type 1-1, mixed (tersoff1989.cu:115-140)
  int version = 0; // 3, 4, 5
  // nep4[_zbl]_temperature (nep.cu:125-130, model_type 3): the ANN has one more input, the temperature handed to
  // NEP::compute(temperature, ...).  `dim` below stays the DESCRIPTOR dimension (what the kernels see); the extra input's
  // weights and scaler are kept aside and folded into the hidden-layer bias whenever the temperature changes
  // (EngineT::set_temperature): tanh(sum_d w0[j][d] q[d] + w0[j][dim] * T * q_scaler[dim] - b0[j]).
  bool temperature_model = false;
  std::vector<float> w0_temp; // [T][neuron]: the last column of the file's w0
  float q_scaler_temp = 0.0f; // the last entry of the file's q_scaler
  bool zbl_enabled = false, zbl_flexible = false;
  double zbl_rc_inner = 0, zbl_rc_outer = 0;
  bool zbl_typewise = false;            // zbl <rc_inner> <rc_outer> <factor> (nep.cu:183-186)
  double zbl_typewise_factor = 0;
  std::vector<float> zbl_rc_outer_pair; // [T*T] min((R_cov(Z1) + R_cov(Z2)) * factor, rc_outer), when zbl_typewise
  int num_types = 0;
  std::vector<std::string> symbols;
  std::vector<int> atomic_numbers;
  std::vector<double> rc_radial, rc_angular; // per type
  double rc_radial_max = 0, rc_angular_max = 0;
  int MN_radial = 0, MN_angular = 0; // enlarged by 1.25 (nep.cu:234-235)
  int n_max_radial = 0, n_max_angular = 0, basis_size_radial = 0, basis_size_angular = 0;
  int L_max = 0, has_q_222 = 0, has_q_1111 = 0, num_L = 0, dim = 0, num_neurons = 0;
  int has_q_112 = 0, has_q_123 = 0, has_q_233 = 0, has_q_134 = 0; // optional trailing l_max flags (nep.cu:275-310)
  int num_para_ann = 0, num_para = 0, num_c_radial = 0;

  // raw parameters in file order (ANN, descriptor c, then q_scaler[dim])
  std::vector<double> params;
  std::vector<double> zbl_para; // 10 per unordered type pair (flexible ZBL only)

  // --- re-laid-out single-precision tables (what the device consumes) ---
  // c_rad[(t1*T+t2)][n][k], c_ang likewise  (nep.cu:75-98 re-layout)
  std::vector<float> c_rad, c_ang;
  // per type: w0[neuron][dim], b0[neuron], w1[neuron], b1t (nep5 per-type output bias, else 0)
  std::vector<float> w0, b0, w1, b1t;
  float b1 = 0.0f;
  std::vector<float> q_scaler;
  std::vector<float> rc_radial_f, rc_angular_f;
  std::vector<float> zbl_para_f;

  // --- a model served by the kernels of a LARGER compiled shape (embed_model below) ---
  // dmap[d of the file's descriptor] = component of the padded descriptor (identity, empty, for a model that is not embedded);
  // file_* keep the header as the file states it (nepmi_model_info reports the model, not the kernels that serve it)
  std::vector<int> dmap;
  int file_n_max_radial = 0, file_n_max_angular = 0, file_basis_size_radial = 0, file_basis_size_angular = 0;
  int file_L_max = 0, file_has_q_222 = 0, file_has_q_1111 = 0, file_num_L = 0, file_dim = 0;
  bool embedded() const { return !dmap.empty(); }
};

// returns empty string on success, else the error text; *unsupported is set when the file is a
// valid NEP model that this engine does not cover (charge/dipole variants, extra invariants...).
std::string load_nep_model(const std::string& path, NepModel& m, bool* unsupported);

// Zero padding into a larger shape: n_max -> NR / NA, basis_size -> KR / KA, invariant rows -> L = 1..4, 222, 1111.  The
// padded radial functions have zero coefficients (their descriptor components are identically zero), the padded basis
// functions multiply zero coefficients, and the ANN's weights on every padded component are zero -- so energies, forces
// and virials are those of the model itself (sums of exact zeros added), evaluated by kernels compiled for the larger
// shape: what a model of a shape nobody compiled kernels for gets instead of the run-time-shape kernels (capi_impl.h).
// Needs: a NEP model with l_max_3body <= 4 and none of the optional rows 112 / 123 / 233 / 134, n_max <= NR / NA,
// basis_size <= KR / KA.  Returns false (m untouched) otherwise.
bool embed_model(NepModel& m, int NR, int KR, int NA, int KA);

} // namespace nepmi
