/* intentionally empty: satisfies an #include of the reference device code when it is compiled under oracle/simt/cuda_on_cpu.h (test infrastructure) */
