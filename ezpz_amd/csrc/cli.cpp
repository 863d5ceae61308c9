// `ezpz-amd`: the reference CLI's behaviour (ezpz-cli/src/main.rs) on top of the C ABI.
//   -f / --filepath <path|->   problem file, '-' for stdin            (main.rs:20-27, :225-238)
//   --show-points              print the final values                  (main.rs:33-35, :129-155)
//   -o / --image-path <png>    accepted, but PNG rendering is out of scope here (visualize.rs)
//   --cold                     extension: drop the topology cache before every timed solve
// Protocol (main.rs:81-103): parse, lower, solve once; lower again and solve 100 more times; report the mean
// over those 100 of the whole elapsed time.  Output lines follow main.rs:106-203.
#include <chrono>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/ezpz_amd.h"

static const int NUM_ITERS_BENCHMARK = 100;  // main.rs:18

static const char* kKindNames[] = {
    "LineTangentToCircle", "CircleTangentToCircle", "Distance", "DistanceVar", "VerticalDistance", "HorizontalDistance",
    "Vertical", "Horizontal", "LinesAtAngle", "Fixed", "ScalarEqual", "PointsCoincident", "CircleRadius",
    "LinesEqualLength", "ArcRadius", "Arc", "Midpoint", "PointLineDistance", "VerticalPointLineDistance",
    "HorizontalPointLineDistance", "Symmetric", "PointArcCoincident", "ArcLength", "ArcAngle", "PointsAtAngle"};

static const char* warning_text(const EzpzWarning& w) {  // warnings.rs:62-83
    switch (w.content) {
    case EZPZ_WARN_DEGENERATE:
        return "This geometry is degenerate, meaning two points are so close together that they practically overlap. "
               "This is probably unintentional, you probably should place your initial guesses further apart or choose "
               "different constraints.";
    case EZPZ_WARN_SHOULD_BE_PARALLEL:
        return "Instead of constraining to this angle, constrain to Parallel";
    default:
        return "Instead of constraining to this angle, constraint to Perpendicular";
    }
}

static void print_problem_size(size_t num_vars, size_t num_eqs) {  // main.rs:194-203
    std::printf("Problem size: %zu rows, %zu vars\n", num_eqs, num_vars);
}

int main(int argc, char** argv) {
    std::string filepath;
    bool show_points = false, cold = false;
    for (int i = 1; i < argc; ++i) {
        std::string a = argv[i];
        if ((a == "-f" || a == "--filepath") && i + 1 < argc)
            filepath = argv[++i];
        else if (a.rfind("--filepath=", 0) == 0)
            filepath = a.substr(11);
        else if (a == "--show-points")
            show_points = true;
        else if (a == "--cold")
            cold = true;
        else if ((a == "-o" || a == "--image-path") && i + 1 < argc) {
            ++i;
            std::fprintf(stderr, "note: PNG output is not part of this build (presentation layer is out of scope)\n");
        } else if (a == "-h" || a == "--help") {
            std::printf("Usage: ezpz-amd --filepath <FILEPATH> [--show-points] [--cold]\n");
            return 0;
        } else {
            std::fprintf(stderr, "error: unexpected argument '%s'\n", a.c_str());
            return 2;
        }
    }
    if (filepath.empty()) {
        std::fprintf(stderr, "error: the following required arguments were not provided:\n  --filepath <FILEPATH>\n");
        return 2;
    }
    std::string text;
    if (filepath == "-") {
        std::stringstream ss;
        ss << std::cin.rdbuf();
        text = ss.str();
    } else {
        std::ifstream f(filepath);
        if (!f) {
            std::fprintf(stderr, "Error: No such file or directory (os error 2)\n");
            return 1;
        }
        std::stringstream ss;
        ss << f.rdbuf();
        text = ss.str();
    }
    char err[512] = {0};
    EzpzProblem* parsed = nullptr;
    auto now = std::chrono::steady_clock::now();
    int rc = ezpz_problem_parse(text.data(), text.size(), &parsed, err, sizeof(err));
    if (rc != EZPZ_OK) {
        std::fprintf(stderr, "Error: %s\n", err[0] ? err : ezpz_error_string(rc));
        return 1;
    }
    const size_t n_cs = ezpz_problem_num_constraints(parsed), n = ezpz_problem_num_vars(parsed);
    const EzpzConstraint* cs = ezpz_problem_constraints(parsed);
    const double* guesses = ezpz_problem_guesses(parsed);
    std::vector<uint32_t> ids(n);
    for (size_t i = 0; i < n; ++i) ids[i] = (uint32_t)i;
    std::vector<double> x(n ? n : 1);
    std::vector<uint64_t> unsat(n_cs + 1);
    std::vector<EzpzWarning> warns(4096);
    EzpzConfig cfg;
    ezpz_default_config(&cfg);
    EzpzOutcome out;
    rc = ezpz_solve(cs, n_cs, ids.data(), guesses, n, &cfg, x.data(), unsat.data(), warns.data(), warns.size(), &out);
    size_t nw = out.n_warnings < warns.size() ? (size_t)out.n_warnings : warns.size();
    auto print_warnings = [&]() {  // main.rs:172-180
        if (nw) {
            std::printf("Warnings:\n");
            for (size_t i = 0; i < nw; ++i) std::printf("\t%s\n", warning_text(warns[i]));
        }
    };
    if (rc != EZPZ_OK) {  // print_failure_output, main.rs:205-223
        print_warnings();
        print_problem_size(out.num_vars, out.num_eqs);
        std::fprintf(stderr, "Could not solve system: %s\n", ezpz_error_string(rc));
        if (out.num_eqs > out.num_vars)
            std::fprintf(stderr, "Your system might be overconstrained. Try removing constraints.\n");
        else
            std::fprintf(stderr, "You might have contradictory constraints.\n");
        return 1;
    }
    // It succeeded. Benchmark its perf (main.rs:93-100): a second lowering, then 100 solves.
    EzpzProblem* again = nullptr;
    ezpz_problem_parse(text.data(), text.size(), &again, err, sizeof(err));
    std::vector<double> x2(n ? n : 1);
    std::vector<uint64_t> unsat2(n_cs + 1);
    EzpzOutcome o2;
    auto loop_start = std::chrono::steady_clock::now();
    for (int i = 0; i < NUM_ITERS_BENCHMARK; ++i) {
        if (cold) ezpz_cache_clear();
        ezpz_solve(ezpz_problem_constraints(again), n_cs, ids.data(), ezpz_problem_guesses(again), n, &cfg, x2.data(),
                   unsat2.data(), nullptr, 0, &o2);
    }
    auto loop_end = std::chrono::steady_clock::now();
    auto elapsed = loop_end - now;
    const double steady_us = std::chrono::duration<double, std::micro>(loop_end - loop_start).count() / NUM_ITERS_BENCHMARK;
    const double first_ms = std::chrono::duration<double, std::milli>(loop_start - now).count();
    const long long micros =
        std::chrono::duration_cast<std::chrono::microseconds>(elapsed).count() / NUM_ITERS_BENCHMARK;

    print_warnings();
    if (out.n_unsatisfied) {  // main.rs:182-192
        std::printf("Not all constraints were satisfied:\n");
        for (uint64_t i = 0; i < out.n_unsatisfied; ++i) {
            const EzpzConstraint& c = cs[unsat[i]];
            std::printf("\t%llu: %s\n", (unsigned long long)unsat[i], c.kind < 25 ? kKindNames[c.kind] : "?");
        }
    }
    print_problem_size(out.num_vars, out.num_eqs);
    std::printf("Iterations needed: %llu\n", (unsigned long long)out.iterations);
    std::printf("Solved up to priority: %u\n", out.priority_solved);
    if (!out.converged) std::printf("Error: solver did not converge!\n");
    std::printf("Solved in %lldμs (mean over %d iterations)\n", micros, NUM_ITERS_BENCHMARK);
    std::printf("i.e. %lld solves per second\n", micros > 0 ? 1000000LL / micros : 0LL);
    // extension: the reference's mean includes the first solve, which here carries the one-time GPU context and
    // code-object load; the loop alone is the per-solve cost a long-lived process sees.
    std::printf("Steady state: %.1fμs per solve, i.e. %.0f solves per second (first solve incl. device init: %.1f ms)\n",
                steady_us, steady_us > 0 ? 1e6 / steady_us : 0.0, first_ms);
    if (show_points) {  // main.rs:129-155, label order executor.rs:525-566
        const size_t np = ezpz_problem_num_labels(parsed, 0), nc = ezpz_problem_num_labels(parsed, 1),
                     na = ezpz_problem_num_labels(parsed, 2);
        std::printf("Points:\n");
        for (size_t i = 0; i < np; ++i)
            std::printf("\t%s: (%.2f, %.2f)\n", ezpz_problem_label(parsed, 0, i), x[2 * i], x[2 * i + 1]);
        if (nc) {
            std::printf("Circles:\n");
            for (size_t i = 0; i < nc; ++i) {
                size_t s = 2 * np + 3 * i;
                std::printf("\t%s: center = (%.2f, %.2f), radius = %.2f\n", ezpz_problem_label(parsed, 1, i), x[s], x[s + 1],
                            x[s + 2]);
            }
        }
        if (na) {
            std::printf("Arcs:\n");
            for (size_t i = 0; i < na; ++i) {
                size_t s = 2 * np + 3 * nc + 6 * i;
                std::printf("\t%s: center = (%.2f, %.2f), a = (%.2f, %.2f), b = (%.2f, %.2f)\n",
                            ezpz_problem_label(parsed, 2, i), x[s + 4], x[s + 5], x[s], x[s + 1], x[s + 2], x[s + 3]);
            }
        }
    }
    ezpz_problem_destroy(parsed);
    ezpz_problem_destroy(again);
    return 0;
}
