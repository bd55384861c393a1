from lcm.nn.schedulers.ddim import DDIMScheduler, DDIMSchedulerConfig, DDIMSchedulerOutput  # noqa: F401
