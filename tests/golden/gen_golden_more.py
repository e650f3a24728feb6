"""Second part of gen_golden.py (PSO / NelderMead / BFGS / LM / tinyqr goldens).
Executed by gen_golden.py with `run` and `write` injected."""


def main():
    # G4 — PSO (nlsolver.h:2496-2742). Vanilla is only pinned where it is defined
    # behaviour (particles <= D).
    g4 = {
        "accel_2d_x0_3_3": run("pso", "accelerated", 2, 10, 50, 0, 1000, "3,3", 2),
        "accel_256d_64p": run("pso", "accelerated", 256, 64, 5, 0, 1000, "0.3", 1),
        "accel_8d_bounded": run("pso", "accelerated", 8, 16, 20, 0, 1000, "2.0", 1, 1, -1.5, 1.5),
        "accel_2d_default_stops": run("pso", "accelerated", 2, 10, 5000, 10e-4, 50, "3,3", 0),
        "accel_2d_other_coefs": run("pso", "accelerated", 2, 12, 30, 0, 1000, "3,3", 1, 0, 0, 0,
                                    0.7, 1.5, 1.2),
        "vanilla_16d_10p": run("pso", "vanilla", 16, 10, 20, 0, 1000, "2.0", 1),
        "vanilla_16d_10p_bounded": run("pso", "vanilla", 16, 10, 20, 0, 1000, "2.0", 1, 1, -1.0, 3.0),
    }
    write("pso.json", g4)

    # G6 — BFGS + More-Thuente on the diag+rank-1 quadratic (nlsolver.h:3169-3286,
    # 1527-1891), analytic gradient functor; every objective value is recorded.
    # args: n max_iter grad_eps alpha x0 x0_step trace
    g6 = {
        "n8": run("bfgs", 8, 100, 1e-10, 1, 1, 0, 1),
        "n64": run("bfgs", 64, 100, 1e-10, 1, 1, 0, 1),
        "n1024": run("bfgs", 1024, 100, 1e-10, 1, 1, 0, 1),
        "n64_default_stop": run("bfgs", 64, 100, 5e-3, 1, 1, 0, 1),
        "n100_ragged_start": run("bfgs", 100, 100, 1e-10, 1, 0.5, 0.03, 1),
        "n130_alpha_half": run("bfgs", 130, 100, 1e-10, 0.5, -2, 0.01, 1),
        "n256_max_iter_5": run("bfgs", 256, 5, 0.0, 1, 1, 0, 1),
        "hess_update_3x3": run("hess-update"),
    }
    write("bfgs.json", g6)

    # BFGS with the DEFAULT gradient (fin_diff = finite_difference_gradient<.,.,1>,
    # nlsolver.h:1385-1413, 2849-2855) on built-in objectives; all objective values counted.
    # args: objective(0 Rosenbrock chain, 1 sphere, 2 Styblinski-Tang) n max_iter grad_eps alpha
    #       x0 x0_step trace_cap
    g6fd = {
        "rosenbrock_n2": run("bfgs-fd", 0, 2, 100, 1e-8, 1, -1.2, 2.2, 64),
        "rosenbrock_n4": run("bfgs-fd", 0, 4, 100, 1e-8, 1, -1.2, 0.7, 64),
        "rosenbrock_n16_default_stop": run("bfgs-fd", 0, 16, 100, 5e-3, 1, 0.8, 0.02, 64),
        "rosenbrock_n128_20iters": run("bfgs-fd", 0, 128, 20, 0.0, 1, 0.9, 0.001, 64),
        "sphere_n5": run("bfgs-fd", 1, 5, 100, 1e-10, 1, 3, -0.5, 64),
        "sphere_n130_alpha_half": run("bfgs-fd", 1, 130, 30, 1e-10, 0.5, 1, 0.01, 64),
        "styblinski_tang_n8": run("bfgs-fd", 2, 8, 100, 1e-8, 1, -2.5, 0.1, 64),
    }
    write("bfgs_fd.json", g6fd)

    # LevenbergMarquardt with its DEFAULT functors (fin_diff, fin_diff_h: nlsolver.h:1385-1517,
    # 3428-3545) on built-in objectives; all objective values counted.
    # args: objective n max_iter lambda f_delta x0 x0_step trace_cap
    g8fd = {
        "rosenbrock_n2_example_start": run("lm-fd", 0, 2, 100, 10, 1e-12, 2, 5, 64),
        "rosenbrock_n4_near_minimum": run("lm-fd", 0, 4, 30, 10, 1e-12, 0.9, 0.02, 64),
        "rosenbrock_n4_indefinite_nan": run("lm-fd", 0, 4, 20, 10, 1e-12, -1.2, 0.7, 64),
        "rosenbrock_n16_6iters": run("lm-fd", 0, 16, 6, 10, 0, 0.95, 0.002, 64),
        "sphere_n5": run("lm-fd", 1, 5, 50, 10, 1e-12, 3, -0.5, 64),
        "styblinski_tang_n8": run("lm-fd", 2, 8, 30, 10, 1e-12, -2.5, 0.1, 64),
        "sphere_n64_3iters_lambda1": run("lm-fd", 1, 64, 3, 1, 0, 1, 0.01, 64),
        # past 64 parameters (the device's workgroup-per-problem kernels in reference order)
        "rosenbrock_n100_2iters": run("lm-fd", 0, 100, 2, 10, 0, 0.95, 0.0005, 64),
        "styblinski_tang_n130_2iters": run("lm-fd", 2, 130, 2, 10, 0, -2.5, 0.01, 64),
    }
    write("lm_fd.json", g8fd)

    # N4 — SANN (nlsolver.h:2744-2815) with the reference's xorshift generator.
    # args: objective n max_iter temp_iter temp_max x0 x0_step minimize trace_cap
    gsann = {
        "rosenbrock_n2_default_schedule": run("sann", 0, 2, 200, 10, 10, 2, 5, 1, 64),
        "rosenbrock_n8": run("sann", 0, 8, 300, 10, 10, 0.5, 0.1, 1, 64),
        "sphere_n16_hot": run("sann", 1, 16, 100, 5, 50, 3, -0.25, 1, 64),
        "styblinski_tang_n6": run("sann", 2, 6, 400, 10, 10, -1, 0.5, 1, 64),
        "sphere_n4_maximize": run("sann", 1, 4, 60, 10, 10, 1, 0.5, 0, 64),
        "rosenbrock_n130_ragged": run("sann", 0, 130, 40, 10, 10, 0.4, 0.001, 1, 64),
        "temp_iter_1_no_moves": run("sann", 0, 4, 25, 1, 10, 0.3, 0.2, 1, 64),
    }
    write("sann.json", gsann)

    # N4 — NelderMeadPSO (nlsolver.h:3546-3920), unbounded overloads. Even n only: init writes one
    # element past an n-vector (3713-3716) and the process aborts for odd n (as NelderMead, B1).
    # args: objective n max_iter eps no_change x0 x0_step minimize trace_cap
    ghyb = {
        "rosenbrock_n2_defaults": run("nmpso", 0, 2, 1000, 1e-6, 20, 2, 5, 1, 64),
        "rosenbrock_n4": run("nmpso", 0, 4, 100, 1e-6, 20, 0.5, 0.1, 1, 64),
        "rosenbrock_n8_200iters": run("nmpso", 0, 8, 200, 0, 1000, 0.5, 0.1, 1, 64),
        "rosenbrock_n16": run("nmpso", 0, 16, 100, 1e-6, 20, 0.5, 0.1, 1, 64),
        "sphere_n6": run("nmpso", 1, 6, 150, 1e-9, 20, 3, -0.4, 1, 64),
        "styblinski_tang_n4_maximize": run("nmpso", 2, 4, 60, 0, 1000, 0.5, 0.3, 0, 64),
        "rosenbrock_n130_ragged": run("nmpso", 0, 130, 30, 0, 1000, 0.4, 0.001, 1, 64),
    }
    write("nmpso.json", ghyb)

    # G8/G9 — LevenbergMarquardt (nlsolver.h:3428-3545) with Gauss-Newton functors, its
    # Cholesky solve (251-330) and tinyqr (291-310, 437-470).
    seed = 12374563468
    g8 = {
        "exp_default": run("lm-exp"),
        "exp_lambda1_5iters": run("lm-exp", 1, 5, 0),
        # lm-tanh seed problem m n lambda max_iter f_delta
        "tanh_m16_n4": run("lm-tanh", seed, 0, 16, 4, 10, 20, 0),
        "tanh_m64_n8_p3": run("lm-tanh", seed, 3, 64, 8, 10, 20, 0),
        "tanh_m128_n64": run("lm-tanh", seed, 1, 128, 64, 10, 12, 0),
        "tanh_m512_n64": run("lm-tanh", seed, 0, 512, 64, 10, 20, 0),
        "tanh_m512_n64_default_stop": run("lm-tanh", seed, 5, 512, 64, 10, 100, 1e-12),
        "tinyqr_example": run("tinyqr-example"),
        "linalg_n4": run("linalg", 4, 1, 0.5),
        "linalg_n16": run("linalg", 16, 2, 10),
        "linalg_n64": run("linalg", 64, 3, 10),
    }
    write("lm.json", g8)

    # G9b — tinyqr on rectangular systems (n >= p; a25): qr_decomposition + lm on pseudo-random
    # X, y: tinyqr-rect n p seed
    write("tinyqr.json", {f"rect_{n}x{p}": run("tinyqr-rect", n, p, sd) for n, p, sd in
                          [(1, 1, 1), (3, 1, 2), (5, 3, 3), (12, 5, 4), (8, 8, 5), (65, 64, 6),
                           (200, 17, 7), (576, 64, 8)]})

    # G5 — NelderMead (nlsolver.h:2099-2300) on Rosenbrock-ND. Even D only: the reference's
    # simplex ctor writes one element past an n-vector, which for odd n lands on
    # the allocator's chunk header and aborts the process (observed for D = 3).
    # args: D step max_iter eps no_change restarts x0 x0_step bounded upper lower minimize trace
    g5 = {
        "example_2d": run("nm", 2, -1, 500, 1e-6, 20, 0, 2, 5, 0, 0, 0, 1, 1),
        "d4_200iters": run("nm", 4, -1, 200, 0, 1000, 0, 0.5, 0, 0, 0, 0, 1, 1),
        "d4_fixed_step": run("nm", 4, 0.75, 120, 0, 1000, 0, -1.2, 0.3, 0, 0, 0, 1, 1),
        "d16_bounded": run("nm", 16, -1, 300, 0, 1000, 0, 0.5, 0.05, 1, 1.5, -1.0, 1, 0),
        "d8_restarts": run("nm", 8, -1, 150, 1e-6, 20, 2, 0.5, 0.1, 0, 0, 0, 1, 0),
        "d6_maximize_bounded": run("nm", 6, -1, 100, 0, 1000, 0, 0.3, 0.2, 1, 2.0, -2.0, 0, 1),
        "d128_2000iters": run("nm", 128, -1, 2000, 0, 100000, 0, 0.5, 0, 0, 0, 0, 1, 0),
        "d130_ragged": run("nm", 130, -1, 600, 0, 100000, 0, 0.4, 0.001, 0, 0, 0, 1, 0),
        "simplex_init_1234": run("simplex-init"),
    }
    write("nm.json", g5)
